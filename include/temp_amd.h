/*
 * temp_amd.h -- C ABI of the MI355X-native TeMP snapshot-encoder hot path (libtemp_amd.so).
 *
 * The TeMP reference (JiapengWu/TeMP) is pure Python and has no plugin / FFI layer; the seam this
 * library plugs into is the `ent_encoder` object installed by `build_model()`
 * (models/TKG_Module.py:37, models/DynamicRGCN.py:32-33, models/BiDynamicRGCN.py:14-15,
 * baselines/StaticRGCN.py:14-18).  Each entry point below names the reference code it replaces.
 * The Python mirror of the reference interface (package temp_amd) binds these symbols with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - All arithmetic is fp32; all indices are int32; all matrices are dense row-major.
 *   - Every pointer is a DEVICE pointer unless stated otherwise.  The caller owns every buffer,
 *     including workspaces; the library never allocates or frees device memory and never synchronises
 *     (temp_rgcn_bwd / temp_rgcn_table_bwd order one side-stream branch against the caller's stream with two
 *     events, TEMP_OPT_OVERLAP: all of its work is complete, in stream order, when the call's last launch is).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  The device is whatever
 *     is current in the calling thread.
 *   - Return value: 0 on success, a TEMP_E_* code otherwise (temp_error_string() describes it).
 *     Nothing is thrown across the ABI.  All functions are re-entrant (no global mutable state
 *     apart from the explicit switches of temp_set_option() and the bench-only trace),
 *     so forward may run on one thread and backward on PyTorch's autograd thread.
 *   - "nullable" arguments may be NULL.
 *   - Arithmetic of the dense products (TEMP_OPT_MFMA_BF16X3 = 1, the default): a product C = A . B whose row count (summed
 *     over the problems of one launch) is >= 16 384 runs on the bf16 matrix pipe as six products of an exact three-way split of
 *     both fp32 operands (fp32-equivalent accuracy: the dropped terms are below one fp32 rounding per product); below that
 *     row count the same call runs on the fp32 MFMA instructions.  The two agree to fp32 rounding, not bit for bit, so the
 *     same layer is bit-different on either side of the 16 384-row line; within one shape every result is bitwise repeatable.
 *     Non-finite inputs: the fp32 kernels propagate them as IEEE arithmetic does.  In the split kernels a non-finite operand
 *     (either side) makes every output that depends on it NON-FINITE -- NaN where fp32 arithmetic may give +-inf: the infinite
 *     piece of the split meets zero pieces of the other operand (inf . 0) -- and leaves all other outputs bit-identical.
 *     Round 6 (TEMP_OPT_MFMA_F16X2 = 1, the default): where an operand arrives with MAGNITUDE KEYS (the *_keys entry points below;
 *     a key = the fp32 bits of a row's / column's largest magnitude with the sign cleared) -- or the product is wide or deep enough
 *     to pay for taking them (N >= 512 or K >= 400) -- the same product runs on the f16 matrix pipe as THREE products of a two-way
 *     split of operands scaled by powers of two (csrc/split_f16.hpp).  Same accuracy class (<= 1e-6 of sum |a||b| against fp64,
 *     tests/test_gpu_f16_split.py), same non-finite contract; an element more than 2^17 below its row's (column's) largest keeps an
 *     ABSOLUTE error of 2^-39 of that largest value instead of a relative one.  Keys handed in must BOUND their row / column (a
 *     key that is too small overflows f16: the outputs of that row / column become non-finite).
 */
#ifndef TEMP_AMD_H
#define TEMP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEMP_ABI_VERSION 2

enum {
  TEMP_OK = 0,
  TEMP_E_BADARG = 1,      /* NULL / negative / inconsistent argument                          */
  TEMP_E_UNSUPPORTED = 2, /* shape outside what the kernels implement (see each function)    */
  TEMP_E_WORKSPACE = 3,   /* workspace too small                                             */
  TEMP_E_LAUNCH = 4       /* hipLaunch / hipGetLastError failure                             */
};

enum { TEMP_ACT_NONE = 0, TEMP_ACT_RELU = 1 };
enum { TEMP_GRU_TORCH = 0,   /* nn.GRU single step, gates r,z,n          (models/RRGCN.py:72,84)   */
       TEMP_GRU_TYPE1 = 1 }; /* "type-1" GRUCell, r,z from hidden only   (models/GRU_cell.py:7-31) */

int temp_abi_version(void);
const char* temp_error_string(int code);

/* Process-wide kernel-selection switches (A/B runs and bit-comparisons in ONE process; every setting computes the
 * same result to fp32 accuracy).  Each is one relaxed atomic int, read at launch time: a set is seen by every launch
 * issued after it returns, on any thread.  Defaults below; the environment variable named on each line, read ONCE
 * when the library is loaded, overrides the default (kept for runs that cannot call into the library first).
 * temp_set_option returns the previous value, or -1 for an unknown key; temp_get_option -1 for an unknown key. */
enum {
  TEMP_OPT_MFMA_BF16X3 = 0, /* 1: large fp32 products on the 16-bit matrix pipes through an operand split with fp32-equivalent accuracy
                               (which split: TEMP_OPT_MFMA_F16X2);  0: every product on the fp32 MFMA kernels
                                                                                           [TEMP_MFMA=f32 -> 0]       default 1 */
  TEMP_OPT_TN_SPLIT = 1,    /* 1: weight-gradient blocks of 4 row tiles x (4+3) column tiles  [TEMP_TN_SPLIT=0 -> 0]  default 1 */
  TEMP_OPT_RGCN_SCALAR = 2, /* 1: wide-row edge kernels keep per-edge quantities in SGPRs     [TEMP_RGCN_SCALAR=0 -> 0] default 1 */
  TEMP_OPT_GEMM_STREAM = 3, /* 1: force the streaming row-panel GEMM instead of weights-resident [TEMP_GEMM_STREAM=1 -> 1] default 0 */
  TEMP_OPT_GRU_STREAM = 4,  /* 1: force the streaming GRU cell kernel                          [TEMP_GRU_STREAM=1 -> 1] default 0 */
  TEMP_OPT_RGCN_TILE = 5,   /* 1: aggregation and d/dh stage a member snapshot's rows in LDS when the graph carries member tables
                               2: the weight-gradient kernel too (slower at the measured shapes)
                               3: the weight-gradient kernel with the gradient rows in LDS and the x rows read through L2 (same
                                  bits as 2; no faster than the gather kernel at the measured shapes)
                               0: always gather through L2 (bit-identical results)             [TEMP_RGCN_TILE=<n>]    default 1 */
  TEMP_OPT_DEBUG = 6,       /* development ablations inside instrumented kernels; 0 (off) in every product run          default 0 */
  TEMP_OPT_OVERLAP = 7,     /* 1: a layer's backward launches the relation-weight gradient -- and, without dropout, the loop-weight
                               and bias gradients behind it -- on a library-owned side stream (fork / join events on the caller's
                               stream: a parallel branch under HIP-graph capture) so that they overlap the d/dh aggregation and
                               the self-loop product; 0: everything on the caller's stream
                                                                                               [TEMP_OVERLAP=0 -> 0]    default 1 */
  TEMP_OPT_GEMM_RESIDENT = 8, /* 1: large fp32 products with K <= 208 keep the packed weights of four column tiles resident in LDS and
                               stream row panels through them (gemm_bxr.hpp); 0: one row tile per block, weights staged per slab
                                                                                               [TEMP_GEMM_RESIDENT=0 -> 0] default 1 */
  TEMP_OPT_MFMA_F16X2 = 9,  /* (with TEMP_OPT_MFMA_BF16X3 = 1)  1: where a kernel exists, three f16 MFMA products of the scaled two-way
                               split (split_f16.hpp: per-row / per-column power-of-two scales, residual <= one fp32 rounding);
                               0: six bf16 MFMA products of the exact three-way split everywhere (gemm_bx.hpp)
                                                                                           [TEMP_MFMA=bf16x3 -> 0]    default 1 */
  TEMP_OPT_COUNT = 10
};
int temp_set_option(int key, int value);
int temp_get_option(int key);
/* Diagnostic: how often a launch found no scratch slot for its packed weights / k-slice partials (more than 8 busy streams
 * on one device) and took the scratch-free kernels instead.  0 in every supported configuration. */
long long temp_scratch_refused(void);
/* Diagnostic: launches of f16-split kernels (TEMP_OPT_MFMA_F16X2) so far in this process -- a test's proof that the product it
 * checked did not silently take the six-product bf16 route. */
long long temp_f16_launches(void);
/* Diagnostic: edge-kernel launches (aggregation, d/dh, d/dweight) that took the LDS-tiled path (TempMembers present, member fits). */
long long temp_tile_launches(void);
/* Development only: a device buffer of `words` int64 into which instrumented kernels write cycle-counter stamps (NULL: off).
 * Not used by the product path or the tests. */
void temp_set_debug_buffer(void* device_ptr, size_t words);

/* ------------------------------------------------------------------------------------------------
 * Segmented edge lists.  One batched snapshot graph (the disjoint union `dgl.batch` builds at
 * models/DynamicRGCN.py:92) is handed over as three sorted views of the same E edges, each cut
 * into chunks of at most TEMP_CHUNK edges (TEMP_CHUNK_REL for the by-relation view, whose segments
 * are few and long) that never straddle a segment:
 *
 *   by destination  (forward aggregation,  RGCNLayer.propagate  models/RGCN.py:100-104)
 *   by source       (d/dh of the aggregation)
 *   by relation     (d/dweight of the aggregation, RGCNLayer.msg_func models/RGCN.py:91-98)
 *
 * A view holds, per edge in sorted order, the two "other" attributes (`a`, `b`), and per chunk the
 * segment id (node or relation row), its [beg,end) edge range and a partial-result slot:
 *   slot == -1 : the chunk is the whole segment, its result is final;
 *   slot >= 0  : the segment spans several chunks; this chunk's partial goes to partial slot `slot`
 *                and `fix_*` lists every such segment with its first slot and slot count (slots of
 *                one segment are consecutive and are summed in order => deterministic).
 * ---------------------------------------------------------------------------------------------- */
#define TEMP_CHUNK 64
#define TEMP_CHUNK_REL 128

/* Dropout of the self-loop message (RGCNLayer.forward, models/RGCN.py:57-59; the reference's default --dropout is 0.1).
 * The keep mask is a counter-based hash of (seed, node row, output column), so forward and backward regenerate the
 * same mask and nothing is stored:  loop_message[row, col] *= (hash(seed, row, col) < p) ? 0 : 1 / (1 - p).
 * A NULL TempDropout* (or p == 0) means no dropout. */
typedef struct TempDropout {
  float p;
  uint64_t seed;
} TempDropout;

typedef struct TempEdgeView {
  int32_t n_seg;            /* number of segments (nodes, or relation rows)                        */
  int32_t n_edges;          /* E                                                                   */
  const int32_t* a;         /* [E]  by-dst: src node   | by-src: dst node | by-rel: src node       */
  const int32_t* b;         /* [E]  by-dst: relation   | by-src: relation | by-rel: dst node       */
  int32_t n_chunks;
  const int32_t* chunk_seg; /* [n_chunks]                                                          */
  const int32_t* chunk_beg; /* [n_chunks]                                                          */
  const int32_t* chunk_end; /* [n_chunks]                                                          */
  const int32_t* chunk_slot;/* [n_chunks]                                                          */
  int32_t n_partial;        /* total partial slots                                                 */
  int32_t n_fix;            /* segments that span more than one chunk                              */
  const int32_t* fix_seg;   /* [n_fix]                                                             */
  const int32_t* fix_slot;  /* [n_fix] first slot                                                  */
  const int32_t* fix_cnt;   /* [n_fix] number of slots                                             */
} TempEdgeView;

/* Optional member tables of a batched graph (the disjoint union `dgl.batch` builds, models/DynamicRGCN.py:92): the union's
 * nodes, the edge arrays of every view and the chunk lists of every view are MEMBER-MAJOR (member m owns the node rows
 * [node_off[m], node_off[m+1]), positions [edge_off[m], edge_off[m+1]) of every view's a/b arrays and chunks
 * [chunk_off[v][m], chunk_off[v][m+1]) of view v = 0 by_dst, 1 by_src, 2 by_rel; no edge crosses members).  With them the edge
 * kernels run one workgroup per (member, feature slice) with the member's rows staged in LDS (block-diagonal relation weights
 * make feature slices independent) instead of gathering rows through L2 -- same chunks, same summation order, bit-identical
 * results.  n_members == 0 (or a member too large for LDS): the kernels gather from global memory as before. */
typedef struct TempMembers {
  int32_t n_members;
  int32_t max_nodes;        /* largest member: nodes                                               */
  int32_t max_edges;        /*                 edge-array positions (edge_off[m+1] - edge_off[m])  */
  int32_t max_chunks[3];    /*                 chunks in the by_dst / by_src / by_rel view         */
  const int32_t* node_off;  /* [n_members + 1]  device                                             */
  const int32_t* edge_off;  /* [n_members + 1]  device                                             */
  const int32_t* chunk_off; /* [3][n_members + 1] device                                           */
  const int32_t* fix_off;   /* [2][n_members + 1] device, nullable: first fix-up entry of every member in the by_dst / by_src
                               view (their fix_* lists are member-major).  Present: the LDS-tiled aggregation / d-dh kernels sum
                               a member's multi-chunk segments themselves, after their walk (one launch less per view). */
} TempMembers;

typedef struct TempGraph {
  int32_t n_nodes;          /* sum of nodes over the batched snapshots                             */
  int32_t n_edges;
  const float* nnorm;       /* [n_nodes] 1/in_degree, 0 for in_degree 0 (utils/utils.py:74-79)     */
  const int32_t* in_deg;    /* [n_nodes]                                                           */
  const int32_t* out_deg;   /* [n_nodes]                                                           */
  TempEdgeView by_dst;
  TempEdgeView by_src;      /* needed by temp_rgcn_bwd only                                        */
  TempEdgeView by_rel;      /* needed by temp_rgcn_bwd only; n_seg = number of weight rows (2R)    */
  TempMembers members;      /* optional (n_members = 0: absent)                                    */
} TempGraph;

/* ------------------------------------------------------------------------------------------------
 * RGCN layer  (RGCNLayer.forward, models/RGCN.py:53-76, dropout = 0)
 *
 *   out[v] = act( nnorm[v]^2 * sum_{(u,r) in In(v)} h[u] . BD(W[r])  [+ bias]  +  h[v] . loop_w )
 *
 * BD(W[r]) is the block-diagonal matrix of `num_bases` blocks of (si x so), si = d_in/num_bases,
 * so = d_out/num_bases, stored row-major per block (b*si*so + i*so + o) -- the `.view(-1, si, so)`
 * of models/RGCN.py:92-93.  The double normalisation (edge norm then node norm) is the
 * reference's.  `h_ids` (nullable) fuses the embedding gather of
 * models/DynamicRGCN.py:93: row v of the input is h[h_ids[v]].
 * Supported: d_in == d_out, d_in % 4 == 0, si == so in {1,2,4} (fast path) or any si,so
 * (generic path).  workspace >= temp_rgcn_fwd_workspace().
 * ---------------------------------------------------------------------------------------------- */
size_t temp_rgcn_fwd_workspace(const TempGraph* g, int d_out);
int temp_rgcn_fwd(const TempGraph* g, const float* h, const int32_t* h_ids /*nullable*/,
                  int d_in, int d_out, int num_bases, int n_rel_rows,
                  const float* weight, const float* loop_w, const float* bias /*nullable*/, int act,
                  float* out, void* workspace, size_t workspace_bytes, const TempDropout* drop, void* stream);

/* Backward of the above (autograd of models/RGCN.py:53-104).
 *   d_h      [n_nodes, d_in]   written (not accumulated)
 *   d_weight [n_rel_rows, num_bases*si*so]  written; rows of unused relations are zeroed
 *   d_loop_w [d_in, d_out]     written
 *   d_bias   [d_out]           written (nullable iff bias was NULL)
 * `out` is the forward output (used for the ReLU mask only when act == TEMP_ACT_RELU).
 * `h` must be the materialised [n_nodes, d_in] input (no fused gather in backward). */
size_t temp_rgcn_bwd_workspace(const TempGraph* g, int d_in, int d_out, int num_bases, int n_rel_rows);
int temp_rgcn_bwd(const TempGraph* g, const float* h, const float* out, const float* d_out_grad,
                  int d_in, int d_out, int num_bases, int n_rel_rows,
                  const float* weight, const float* loop_w, int has_bias, int act,
                  float* d_h, float* d_weight, float* d_loop_w, float* d_bias /*nullable*/,
                  void* workspace, size_t workspace_bytes, const TempDropout* drop, void* stream);

/* The two halves of temp_rgcn_bwd, for a layer INSIDE a recurrence (the reference's default flags: both layers recurrent,
 * models/RRGCN.py:179-204): d_h position by position, the weight gradients once over the union of all positions' graphs.
 *   temp_rgcn_bwd_dh:      d_h [n_nodes, d_in] only.  act == TEMP_ACT_RELU: the ReLU-masked gradient is written to dz_out
 *                          [n_nodes, d_out] (required then); with dropout the masked gradient of the self-loop message is written
 *                          to dzm_out (required then).  act == TEMP_ACT_NONE without dropout: both may be NULL (dz = d_out_grad).
 *   temp_rgcn_bwd_weights: d_weight / d_loop_w / d_bias from h [n_nodes, d_in], dz (the masked gradient) and dzm (NULL: = dz) --
 *                          the same kernels temp_rgcn_bwd runs, on whatever graph the rows belong to (a union of positions).
 * Workspace: temp_rgcn_bwd_workspace of the graph. */
int temp_rgcn_bwd_dh(const TempGraph* g, const float* out /*nullable unless relu*/, const float* d_out_grad, int d_in, int d_out, int num_bases,
                     int n_rel_rows, const float* weight, const float* loop_w, int act, float* d_h, float* dz_out /*nullable*/,
                     float* dzm_out /*nullable*/, void* workspace, size_t workspace_bytes, const TempDropout* drop, void* stream);
int temp_rgcn_bwd_weights(const TempGraph* g, const float* h, const float* dz, const float* dzm /*nullable*/, int d_in, int d_out, int num_bases,
                          int n_rel_rows, int has_bias, float* d_weight, float* d_loop_w, float* d_bias /*nullable*/, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Same layer when its input is a ROW GATHER of a table, h = table[ids]  (layer 1 of every TeMP encoder:
 * g.ndata['h'] = ent_embeds[g.ndata['id']], models/DynamicRGCN.py:93).  h is never materialised:
 *   fwd: the aggregation gathers through ids; the self-loop term is (table . W_loop)[ids] -- one N_table-row GEMM
 *        instead of one over every node row of the batch;
 *   bwd: what is linear in the gathered rows is summed per table row first (inv_ptr / inv_order = the ids grouped
 *        by table row, see temp_segment_sum_rows):  d_table = segsum(d_h_agg) + segsum(dz) . W_loop^T  [n_table, d_in]
 *        (fully written, deterministic), d_loop_w = table^T . segsum(dz).
 * Results equal temp_rgcn_fwd/bwd on h = table[ids] followed by the gather's adjoint, up to fp32 summation order.
 * ---------------------------------------------------------------------------------------------- */
size_t temp_rgcn_table_fwd_workspace(const TempGraph* g, int n_table, int d_out);
int temp_rgcn_table_fwd(const TempGraph* g, const float* table, const int32_t* ids, int n_table, int d_in, int d_out, int num_bases,
                        int n_rel_rows, const float* weight, const float* loop_w, const float* bias, int act, float* out, void* workspace,
                        size_t workspace_bytes, const TempDropout* drop, void* stream);
size_t temp_rgcn_table_bwd_workspace(const TempGraph* g, int n_table, int d_in, int d_out, int num_bases);
int temp_rgcn_table_bwd(const TempGraph* g, const float* table, const int32_t* ids, const int32_t* inv_ptr, const int32_t* inv_order, int n_table,
                        const float* out, const float* d_out_grad, int d_in, int d_out, int num_bases, int n_rel_rows, const float* weight,
                        const float* loop_w, int has_bias, int act, float* d_table, float* d_weight, float* d_loop_w, float* d_bias,
                        void* workspace, size_t workspace_bytes, const TempDropout* drop, void* stream);

/* Isolated-entity variant (RGCNLayer.forward_isolated, models/RGCN.py:78-89):
 *   out = act( e + e . loop_w [+ bias] )          e: [n, d]                                    */
int temp_rgcn_isolated_fwd(int n, int d, const float* e, const float* loop_w, const float* bias /*nullable*/,
                           int act, float* out, const TempDropout* drop, void* stream);
size_t temp_rgcn_isolated_bwd_workspace(int n, int d);
int temp_rgcn_isolated_bwd(int n, int d, const float* e, const float* out, const float* d_out_grad,
                           const float* loop_w, int has_bias, int act,
                           float* d_e, float* d_loop_w, float* d_bias /*nullable*/,
                           void* workspace, size_t workspace_bytes, const TempDropout* drop, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Recurrent update: decay of the previous state + one GRU step
 * (GRRGCNLayer.forward, models/RRGCN.py:77-89; BiGRRGCNLayer, models/BiRRGCN.py:27-63).
 *
 *   hdec[i] = prev[prev_idx ? prev_idx[i] : i] * exp(-lambda * dt[i])                (fixed)
 *           = prev[...] * exp(-max(0, decay_w * dt[i] + decay_b))                    (learnable,
 *                                                     RGCNLayer.decay_hidden models/RGCN.py:106-107)
 *   prev_idx[i] == -1  =>  previous state is the zero vector (entity inactive at the previous
 *                          window position: the reference re-zeroes its dense history every step,
 *                          models/DynamicRGCN.py:47-54).
 *   TEMP_GRU_TORCH:  r = sig(W_ir x + b_ir + W_hr hdec + b_hr),  z likewise,
 *                    n = tanh(W_in x + b_in + r * (W_hn hdec + b_hn)),  h' = (1-z) n + z hdec
 *                    w_ih, w_hh: [3d, d] rows ordered r,z,n;  b_ih, b_hh: [3d]
 *   TEMP_GRU_TYPE1:  r = sig(W_hr hdec + b_hr), z = sig(W_hz hdec + b_hz),
 *                    n = tanh(W_in x + b_in + r * (W_hn hdec + b_hn)),  h' = n + z (hdec - n)
 *                    w_ih: [d, d], b_ih: [d];  w_hh: [3d, d], b_hh: [3d]
 *   `learnable`: decay_wb points to 2 DEVICE floats {w, b}; otherwise decay_wb is NULL and
 *   `lambda` is used.
 *   saved: [5, n, d] scratch the backward needs (r, z, n, W_hn hdec + b_hn, hdec).
 * Supported: d % 4 == 0.
 * ---------------------------------------------------------------------------------------------- */
int temp_gru_fwd(int n, int d, int variant,
                 const float* x, const float* prev, const int32_t* prev_idx /*nullable*/, const float* dt,
                 float lambda, const float* decay_wb /*nullable*/,
                 const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                 float* h_out, float* saved, void* stream);

/* Backward.  d_h_out: [n,d] upstream gradient.
 *   d_x    [n,d]  written
 *   d_prev [n,d]  written, in the row order of x (caller scatters through prev_idx)
 *   d_w_ih, d_w_hh, d_b_ih, d_b_hh written
 *   d_decay_wb: 2 device floats written when learnable (nullable otherwise)                       */
size_t temp_gru_bwd_workspace(int n, int d, int variant);
int temp_gru_bwd(int n, int d, int variant,
                 const float* x, const float* prev, const int32_t* prev_idx /*nullable*/, const float* dt,
                 float lambda, const float* decay_wb /*nullable*/,
                 const float* w_ih, const float* w_hh,
                 const float* saved, const float* d_h_out,
                 float* d_x, float* d_prev, float* d_w_ih, float* d_w_hh, float* d_b_ih, float* d_b_hh,
                 float* d_decay_wb /*nullable*/,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Fixed exponential decay as a stand-alone row scale: out[r, :] = x[r, :] * exp(-dt[r] * lambda) -- the recurrent term of the
 * linear-recurrence layers (RRGCNLayer models/RRGCN.py:130-151, BiRRGCNLayer models/BiRRGCN.py:115-140), whose GEMM runs through
 * temp_linear.  Self-adjoint (the backward is the same call on the gradient).  out may alias x. */
int temp_decay_rows(int n, int d, const float* x, const float* dt, float lambda, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Window-batched recurrence.  When only the last layer is recurrent (--rec-only-last-layer,
 * models/RRGCN.py:182-187) the GRU input x of EVERY window position is known before the chain
 * starts, so everything that is not truly sequential is hoisted out of the per-position loop:
 *
 *   temp_gru_input_gates   gi = x . W_ih^T + b_ih for all positions at once  (one MFMA GEMM)
 *   temp_gru_cell_fwd      per position: hdec . W_hh^T on MFMA + gates + blend (reads its gi rows)
 *   temp_gru_cell_bwd      per position: gate gradients + d_prev = (dgh . W_hh + dh*z) * decay;
 *                          the gradient arriving from the NEXT position is gathered through the
 *                          inverse row map `next_idx` (-1 = none) and added to `dh_up` (nullable)
 *   temp_gru_weight_grads  after the chain: d_x = dgi . W_ih, d_W_ih = dgi^T x, d_W_hh = dgh^T hdec,
 *                          bias column sums -- over ALL rows that share one set of GRU weights
 *
 * `saved` planes (r, z, n, W_hn hdec + b_hn, hdec) are `saved_plane` floats apart so that every
 * position writes its rows into one [5, N_total, d] buffer and `hdec` (plane 4) is contiguous for
 * the batched weight gradient.  gi/dgi: [n, 3d] (torch) or [n, d] (type-1); dgh: [n, 3d]; decv: [n].
 * Fixed decay only (learnable decay uses temp_gru_fwd / temp_gru_bwd).
 * ---------------------------------------------------------------------------------------------- */
int temp_gru_input_gates(int n, int d, int variant, const float* x, const float* w_ih, const float* b_ih, float* gi, void* stream);
/* The same for `count` (row block, weight set) pairs of one width in as few launches as possible (four problems each): the input
 * gates of both directions of a bidirectional window chain (models/BiRRGCN.py:206-221: forward_rnn and backward_rnn).  Results
 * are bit-identical to `count` calls of temp_gru_input_gates. */
int temp_gru_input_gates_multi(int count, const int* ns, int d, int variant, const float* const* xs, const float* const* w_ihs,
                               const float* const* b_ihs, float* const* gis, void* stream);
/* ... with the rows of problem i gathered: gis[i] row j = xs[i][x_idx[i][j]] . W_ih^T + b_ih (x_idx: HOST array of device int32
 * tables, an entry may be NULL = rows in order; indices >= 0).  Used to compute the gates of every DISTINCT input row of a
 * window chain once (TempGruChain.gi_index maps chain rows to gi rows).  Within one arithmetic (Conventions: the 16 384-row
 * switch looks at the rows of the whole launch) a row's result does not depend on where it is computed: gates[gi_index[i]] is then bit-identical to
 * temp_gru_input_gates_multi's row i. */
int temp_gru_input_gates_gather_multi(int count, const int* ns, int d, int variant, const float* const* xs, const int32_t* const* x_idx,
                                      const float* const* w_ihs, const float* const* b_ihs, float* const* gis, void* stream);
int temp_gru_cell_fwd(int n, int d, int variant, const float* gi, const float* prev, const int32_t* prev_idx /*nullable*/,
                      const float* dt, float lambda, const float* w_hh, const float* b_hh,
                      float* h_out, float* saved, size_t saved_plane, void* stream);
int temp_gru_cell_bwd(int n, int d, int variant, const float* saved, size_t saved_plane,
                      const float* dh_up /*nullable*/, const float* d_prev_next /*nullable*/, const int32_t* next_idx /*nullable*/,
                      const float* dt, float lambda, const float* w_hh,
                      float* dgi, float* dgh, float* decv, float* d_prev, void* stream);
/* Several independent cells in ONE launch each (forward chain and backward chain of the bidirectional
 * window advance position by position together; count <= 4; arrays are HOST arrays of structs). */
typedef struct TempGruCellFwd {
  int32_t n; const float* gi; const float* prev /* NULL: the cell starts from the zero state (pointwise, no GEMM) */;
  const int32_t* prev_idx /*nullable*/; const float* dt;
  const float* w_hh; const float* b_hh; float* h_out; float* saved /* first row of the cell inside the [5, N, d] planes */;
} TempGruCellFwd;
typedef struct TempGruCellBwd {
  int32_t n; const float* saved; const float* dh_up /*nullable*/; const float* d_prev_next /*nullable*/;
  const int32_t* next_idx /*nullable*/; const float* dt; const float* w_hh;
  float* dgi; float* dgh; float* decv; float* d_prev;
  int32_t no_prev;   /* != 0: the cell started from a zero state (first position of a chain): nothing consumes d_prev, so its
                        GEMM is skipped and d_prev is left holding only the gate kernel's seed */
} TempGruCellBwd;
int temp_gru_cell_fwd_multi(int count, const TempGruCellFwd* cells, int d, int variant, float lambda, size_t saved_plane, void* stream);
int temp_gru_cell_bwd_multi(int count, const TempGruCellBwd* cells, int d, int variant, float lambda, size_t saved_plane, void* stream);
size_t temp_gru_weight_grads_workspace(int n, int d, int variant);
/* hdec may be NULL when every row started from the zero state (d_w_hh = 0, d_b_hh = column sums of dgh). */
int temp_gru_weight_grads(int n, int d, int variant, const float* x, const float* hdec, const float* dgi, const float* dgh,
                          const float* w_ih, float* d_x /*nullable*/, float* d_w_ih, float* d_w_hh, float* d_b_ih, float* d_b_hh,
                          void* workspace, size_t workspace_bytes, void* stream);
/* The same for `count` <= 4 GRUs of one width in ONE weight-gradient launch + ONE reduction + ONE d_x launch (both directions of a
 * bidirectional window chain, models/BiRRGCN.py:206-221): d_w = [2 count, 3d, d] holds d_W_ih_0, d_W_hh_0, d_W_ih_1, ...,
 * d_b = [2 count, 3d] the bias gradients in the same order; xs / hdecs / dgis / dghs / w_ihs / d_xs are HOST arrays of device
 * pointers (a d_xs entry may be NULL).  nn.GRU gate layout only; TEMP_E_UNSUPPORTED (nothing launched) for shapes that need the
 * per-GRU call (type-1 cell, small row counts, a GRU whose rows all start from the zero state: hdecs[i] NULL).  Same sums as the
 * per-GRU call up to the order of the row slices. */
size_t temp_gru_weight_grads_multi_workspace(int count, const int* ns, int d, int variant);
int temp_gru_weight_grads_multi(int count, const int* ns, int d, int variant, const float* const* xs, const float* const* hdecs,
                                const float* const* dgis, const float* const* dghs, const float* const* w_ihs, float* const* d_xs,
                                float* d_w, float* d_b, void* workspace, size_t workspace_bytes, void* stream);

/* Weight / bias gradients and d_x of `count` <= 4 GRUs from their gate-gradient matrices g4s[i] = [ns[i]][4d] (temp_gru_chain_bwd_g4):
 *   d_W_ih = [dr dz dn_i]^T x,  d_W_hh = [dr dz dn_h]^T hdec,  d_b_* = the column sums,  d_x = [dr dz dn_i] . W_ih
 * (the backward of the GRU step, models/RRGCN.py:84) -- ONE weight-gradient launch (k_gru_wgrad: both products of every GRU read
 * their columns of the one matrix through a column map; the row-wise sums take their fragments through LDS transpose reads, no
 * operand is turned in registers), ONE deterministic reduction over the row slices, ONE d_x launch.  Output layout as
 * temp_gru_weight_grads_multi: d_w = [2 count][3d][d] (d_W_ih_0, d_W_hh_0, d_W_ih_1, ...), d_b = [2 count][3d].  xs / hdecs / g4s /
 * w_ihs / d_xs: HOST arrays of device pointers (a d_xs entry may be NULL).  The workspace query returns 0 and the call
 * TEMP_E_UNSUPPORTED (nothing launched) for shapes that take the dgi / dgh calls: d % 8 != 0, d >= 256, fewer than 16 384
 * rows in all, TEMP_OPT_MFMA_BF16X3 off.  fp32-equivalent arithmetic (six bf16 MFMA products of the exact operand split). */
size_t temp_gru_grads_g4_workspace(int count, const int* ns, int d);
int temp_gru_grads_g4(int count, const int* ns, int d, const float* const* xs, const float* const* hdecs, const float* const* g4s,
                      const float* const* w_ihs, float* const* d_xs, float* d_w, float* d_b, void* workspace, size_t workspace_bytes,
                      void* stream);
/* Round 6: the same with the keys temp_gru_chain_bwd_g4_keys hands out (HOST arrays of `count` device pointers; NULL arrays or
 * TEMP_OPT_MFMA_F16X2 = 0: exactly temp_gru_grads_g4).  g4_col_keys[i]: [4d] keys bounding the column magnitudes of g4s[i];
 * g4_row_keys[i]: [ns[i]] keys of the rows' max |[dr dz dn_i]|.  With them the products run as three f16 MFMA products of the
 * scaled two-way split (split_f16.hpp) instead of six bf16 products; hdecs must be decayed GRU states (|hdec| < 4: they are split
 * with the constant scale 2^14).  Same workspace. */
int temp_gru_grads_g4_keys(int count, const int* ns, int d, const float* const* xs, const float* const* hdecs, const float* const* g4s,
                           const float* const* w_ihs, float* const* d_xs, float* d_w, float* d_b, const uint32_t* const* g4_row_keys,
                           const uint32_t* const* g4_col_keys, const uint32_t* const* x_col_keys /* nullable (array or entries): [d] keys
                           bounding the column magnitudes of xs[i], e.g. from temp_gather_rows_keys; else taken here with one pass */,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Persistent window chain: ALL positions of the recurrence in ONE launch per direction of time.
 *
 * The recurrence of the window models is per (window, direction, entity): a row at position p reads only the state of the
 * SAME entity at position p-1 (`prev_idx`; models/DynamicRGCN.py:35-54, models/RRGCN.py:77-89).  So the rows of a chain
 * program split into independent *panels* of <= 32 entity tracks; a workgroup owns a panel, keeps its 32 states in LDS and
 * walks all positions of the chain without ever meeting another workgroup: no grid barrier, no relaunch.  Per position a
 * panel does  hdec . W_hh^T  (32 x 3d x d) on the fp32 MFMA pipe with W_hh streamed from L2 in fragment order
 * (temp_gru_chain_pack), then the gates / blend pointwise; the backward walks the positions in reverse with
 * d_prev = (dgh . W_hh + dh*z) * decay kept in LDS.  Replaces 15 x (temp_gru_cell_fwd_multi) and
 * 15 x (temp_gru_cell_bwd_multi) launches of a seq_len-15 bidirectional window.
 *
 * Tables (device, int32), built once per prepared batch on the host:
 *   panel [n_panels][4] : { GRU index (< n_rnn), first step (row of the step tables), number of steps, 0 }
 *   rows  [S][32]       : per step and track: row in the [N_total, *] buffers | TEMP_CHAIN_HAS_PREV if the track carries a
 *                         state from the step before (else the state is zero), or -1 (track idle at this step)
 *   sinfo [S][4]        : { flags (bit0: some track of the step has a previous state, bit1: write h_out rows),
 *                           up: index into `up` of the upstream gradient block of the step's rows or -1,
 *                           up_row0: first row of that block in the [N_total] row space, 0 }
 * Steps of a panel are consecutive table rows in chain order.  gi / saved / dgi / dgh / h: as for temp_gru_cell_*.
 * Fixed decay only.  d % 4 == 0 and d <= TEMP_CHAIN_MAX_D (LDS), else TEMP_E_UNSUPPORTED.
 * ---------------------------------------------------------------------------------------------- */
#define TEMP_CHAIN_HAS_PREV (1 << 30)
#define TEMP_CHAIN_TRACKS 32
#define TEMP_CHAIN_MAX_RNN 4
#define TEMP_CHAIN_MAX_UP 8
typedef struct TempGruChain {
  int32_t d, variant, n_panels, n_steps;
  int32_t max_steps;                     /* longest panel (<= 64): sizes the per-panel tables the kernels stage in LDS */
  const int32_t* panel; const int32_t* rows; const int32_t* sinfo;
  const float* dt;                       /* [N_total] time gap of every row */
  float lambda;
  size_t saved_plane;
  int32_t n_rnn;
  const float* packed[TEMP_CHAIN_MAX_RNN];   /* temp_gru_chain_pack of each GRU's W_hh */
  const float* b_hh[TEMP_CHAIN_MAX_RNN];
  const int32_t* gi_index;               /* nullable [N_total]: row of `gi` that holds row i's input gates.  Chain rows that read the
                                          * same input row share one gi row (temp_gru_input_gates_gather_multi computes each once);
                                          * NULL: gi has N_total rows, row i's gates in row i.  Forward only: dgi stays per row. */
} TempGruChain;
int temp_gru_chain_supported(int d);
size_t temp_gru_chain_pack_floats(int d);                                   /* floats of one packed W_hh */
int temp_gru_chain_pack(int d, const float* w_hh, float* packed, void* stream);
/* count <= TEMP_CHAIN_MAX_RNN matrices in one launch (w_hh / packed: HOST arrays of device pointers; each packed[i] holds
 * temp_gru_chain_pack_floats(d) floats) */
int temp_gru_chain_pack_multi(int count, int d, const float* const* w_hh, float* const* packed, void* stream);
int temp_gru_chain_fwd(const TempGruChain* c, const float* gi, float* h_out, float* saved, void* stream);
int temp_gru_chain_bwd(const TempGruChain* c, const float* saved, int n_up, const float* const* up /* HOST array of device pointers */,
                       float* dgi, float* dgh, void* stream);
/* The same backward with the gate gradients written ONCE (round 5; nn.GRU gate layout only):
 *   g4 [n_rows][4d] = [dr | dz | dn_i | dn_h]  -- dgi = columns 0 .. 3d-1, dgh = columns 0 .. 2d-1 and 3d .. 4d-1 (two thirds of dgh
 * repeated dgi: 1 600 bytes per row less to write at d = 200).  Bit-identical values.  Consumed by temp_gru_grads_g4. */
int temp_gru_chain_bwd_g4(const TempGruChain* c, const float* saved, int n_up, const float* const* up /* HOST array of device pointers */,
                          float* g4, void* stream);
/* Round 6: the same, also handing out what the consumers of g4 need to split it for the f16 pipe (temp_gru_grads_g4_keys):
 *   row_keys [N_total]    : per row the largest magnitude of [dr dz dn_i] as an unsigned key (the fp32 bits with the sign cleared;
 *                           integer order = magnitude order)
 *   col_keys [n_rnn + n_panels][4d] : rows 0 .. n_rnn - 1 = per GRU and column of g4 a key that BOUNDS the column's largest magnitude
 *                           (integer maxima: order-independent, bit-repeatable); the n_panels rows behind them are scratch (every
 *                           panel's own maxima, reduced by a second small launch); no initialisation needed
 * temp_gru_chain_keys_supported(d): 1 when the chain kernels of this width produce keys (f16 arithmetic selected and its LDS
 * images fit); otherwise this call returns TEMP_E_UNSUPPORTED and the caller uses temp_gru_chain_bwd_g4. */
int temp_gru_chain_keys_supported(int d);
int temp_gru_chain_bwd_g4_keys(const TempGruChain* c, const float* saved, int n_up, const float* const* up, float* g4,
                               uint32_t* row_keys, uint32_t* col_keys, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row gather / scatter helpers of the window loop
 * (get_prev_embeddings / update_time_diff_hist_embeddings / ent_embeds[id],
 *  models/DynamicRGCN.py:35-54,93).
 *   gather:       out[i] = idx[i] >= 0 ? table[idx[i]] : 0                 out: [n, d]
 *   scatter_add:  table[idx[i]] += src[i]  for idx[i] >= 0 (atomic, rows may repeat)
 * ---------------------------------------------------------------------------------------------- */
int temp_gather_rows(int n, int d, const float* table, const int32_t* idx, float* out, void* stream);
/* Magnitude keys (round 6, split_f16.hpp): the key of a value is its fp32 bit pattern with the sign cleared, so integer order is
 * magnitude order and a maximum of keys does not depend on the order it is taken in.  A consumer that runs a product as three f16
 * MFMA products scales every row (activations) / column (a sum over rows) by the power of two its key gives.
 *   temp_absmax_keys      : row_keys[n] (nullable) = key of every row's largest magnitude; col_keys (nullable) = [d] keys bounding
 *                           every column's largest magnitude, followed by temp_keys_cols_size(d) - d words of scratch (the buffer
 *                           holds temp_keys_cols_size(d) words; nothing to initialise).  d % 4 == 0, d <= 256.
 *   temp_gather_rows_keys : temp_gather_rows that also hands out the keys of its OUTPUT rows / columns (same buffers).
 *   temp_linear_keys      : temp_linear with the row keys of A (nullable: taken inside when the product is wide or deep enough to pay
 *                           for the pass, else the six-product bf16 kernels run).
 *   temp_gru_input_gates_gather_multi_keys : x_keys[i] = row keys of xs[i] by source row (nullable array / entries). */
size_t temp_keys_cols_size(int d);
int temp_absmax_keys(int n, int d, const float* x, int ldx, uint32_t* row_keys, uint32_t* col_keys, void* stream);
int temp_gather_rows_keys(int n, int d, const float* table, const int32_t* idx, float* out, uint32_t* row_keys, uint32_t* col_keys, void* stream);
int temp_linear_keys(int M, int N, int K, const float* A, int lda, const uint32_t* a_keys, const float* B, int ldb, int trans_b, float* C, int ldc,
                     void* stream);
int temp_gru_input_gates_gather_multi_keys(int count, const int* ns, int d, int variant, const float* const* xs, const int32_t* const* x_idx,
                                           const uint32_t* const* x_keys, const float* const* w_ihs, const float* const* b_ihs, float* const* gis,
                                           void* stream);
int temp_scatter_add_rows(int n, int d, const float* src, const int32_t* idx, float* table, void* stream);
/* Deterministic adjoint of a gather with a STATIC index list (the ids of a prepared window batch):
 *   out[s] = sum_{j in [seg_ptr[s], seg_ptr[s+1])} src[order[j]]      out: [n_seg, d] fully written (empty segment = 0)
 * seg_ptr [n_seg+1] / order [n_rows] = the gather indices grouped by table row (built once on the host;
 * ids < 0 are left out, so n_rows may be smaller than the gather).  n_rows must be EXACTLY the length of `order`
 * (= seg_ptr[n_seg]): the piece kernels size their partials from it (a smaller value would drop the trailing rows).
 * No atomics; d % 4 == 0, d <= 256.  Tables with very long segments (>= 512 rows per segment on average, e.g. the
 * relation table under the loss) are reduced in two deterministic stages through `workspace`; segmentations of 2-32 rows per
 * segment on average are summed over fixed 32-row pieces of `order` (skew-proof: a hub entity's thousands of rows are many
 * pieces, not one wave's loop) with the piece partials in `workspace`
 * (temp_segment_sum_rows_workspace bytes; 0 for the other shapes; NULL falls back to one wave / one block per segment). */
size_t temp_segment_sum_rows_workspace(int n_seg, int n_rows, int d);
int temp_segment_sum_rows(int n_seg, int n_rows, int d, const int32_t* seg_ptr, const int32_t* order, const float* src, float* out,
                          void* workspace, size_t workspace_bytes, void* stream);
/* The same with the adjoint of a ReLU folded in: out[s][c] = relu_of[s][c] > 0 ? sum : 0, where relu_of [n_seg, d] is the
 * POST-activation table the gather read (y = relu(z) => y > 0 <=> z > 0).  For a gather that is the only consumer of an
 * RGCN layer's ReLU output (models/BiRRGCN.py:202-203 -> the GRU input rows), the layer's backward then takes the gradient as
 * already masked (act = TEMP_ACT_NONE) and the (n, d) mask pass of its own disappears. */
int temp_segment_sum_rows_relu(int n_seg, int n_rows, int d, const int32_t* seg_ptr, const int32_t* order, const float* src, const float* relu_of,
                               float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Plain fp32 MFMA GEMMs (the extra (n,D)@(D,D) terms of the linear-recurrence layers,
 * models/RRGCN.py:141, models/BiRRGCN.py:128-129; the all-entity score matrix below).
 *   temp_linear    C[M,N] = A[M,K] . B          B is [K,N] (trans_b = 0) or stored as [N,K] (trans_b = 1)
 *   temp_linear_tn out[Ka,Nb] = A[M,Ka]^T . B[M,Nb]      (deterministic split over M)
 * K, N, Ka, Nb and all leading dimensions must be multiples of 4.
 * ---------------------------------------------------------------------------------------------- */
int temp_linear(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int trans_b, float* C, int ldc, void* stream);
/* Same product, result stored TRANSPOSED: Ct[N, ldct] with Ct[n, m] = (A . B)[m, n]  (ldct >= M).  For "few rows x many columns"
 * outputs -- a window's ~200 positives scored against 10 000 entities -- the entity axis is made the tall one:
 * scores[P, N_ents] = temp_linear_t(M = N_ents, N = P, A = all_embeds, B = queries (trans_b = 1), Ct = scores). */
int temp_linear_t(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int trans_b, float* Ct, int ldct, void* stream);
/* Several independent products of the same shape class in ONE launch sequence (up to 4 problems per launch):
 *   C_i[M_i, N] = A_i[M_i, K] . B_i        same N, K, leading dimensions and transposition; M_i may differ.
 * The per-window score matrices of the loss (every window scores against its own all-entity table) and their
 * input gradients.  `probs` is a HOST array. */
typedef struct TempLinearProblem { int M; const float* A; const float* B; float* C; } TempLinearProblem;
int temp_linear_multi(int count, const TempLinearProblem* probs, int N, int K, int lda, int ldb, int trans_b, int ldc, void* stream);
size_t temp_linear_tn_workspace(int M, int Ka, int Nb);
int temp_linear_tn(int M, int Ka, int Nb, const float* A, int lda, const float* B, int ldb, float* out, int ldo,
                   void* workspace, size_t workspace_bytes, void* stream);
/* Several weight-gradient-shaped products in one launch sequence (the per-window d_all_b = d_scores_b^T . q_b of the loss,
 * models/TKG_Module.py:202-213 differentiated): C_i[Ka,Nb] = A_i[M_i,Ka]^T . B_i[M_i,Nb], same Ka, Nb and leading dimensions,
 * C_i contiguous (ldc == Nb).  Small M_i run as ONE launch (up to 8 problems, problem index in the grid); shapes of the large-M
 * kernels fall back to one temp_linear_tn per problem.  `probs` is a HOST array; deterministic. */
size_t temp_linear_tn_multi_workspace(int count, int max_m, int Ka, int Nb);
int temp_linear_tn_multi(int count, const TempLinearProblem* probs, int Ka, int Nb, int lda, int ldb, int ldc, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Link-prediction loss (TKG_Module.train_link_prediction, models/TKG_Module.py:202-213;
 * utils/scores.py:4-44).  DistMult and ComplEx are bilinear in (query, candidate), so the scores of
 * the P positives against ALL N entities are ONE GEMM  scores = query[P,D] . all_embeds[N,D]^T
 * (temp_linear, trans_b = 1); the reference instead gathers a (P, 1+neg, D) tensor (1.2 GB at
 * P = 3000, neg = 500).  These two calls do the cross-entropy with label 0 over cand[P,C]
 * (column 0 = the true entity, global ids):
 *   fwd: loss_rows[p] = logsumexp_k scores[p, cand[p,k]] - scores[p, cand[p,0]];  lse_rows saved
 *   bwd: d_scores[P,N] (fully written) = scale[0] * inv_rows * sum_k (softmax_k - [k==0]) e_{cand[p,k]}
 *        (`scale` is a DEVICE float: the upstream gradient of the mean loss)
 * ---------------------------------------------------------------------------------------------- */
int temp_gather_ce_fwd(int P, int C, int N, const float* scores, const int32_t* cand, float* loss_rows, float* lse_rows, void* stream);
int temp_gather_ce_bwd(int P, int C, int N, const float* scores, const int32_t* cand, const float* lse_rows, const float* scale,
                       float inv_rows, const float* row_scale /* nullable [P]: per-row weight instead of inv_rows */, float* d_scores, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Folded query of the bilinear scorers, row gathers fused in (utils/scores.py:4-12 distmult, :26-44 complex;
 * the s / r / o row selections of train_link_prediction, models/TKG_Module.py:202-213):
 *   k = ent_rows[known_idx[p]], r = rel[rel_idx[p]]          (both [*, d] row-major, d % 4 == 0; complex: d % 8 == 0)
 *   q[p] such that  score(p, candidate c) = <q[p], c>;  is_tail[p] != 0: k is the subject and candidates are
 *   objects (mode 'tail'), else k is the object and candidates are subjects (mode 'head'; ignored by distmult).
 *   bwd: per-row gradients d_known_rows[p], d_rel_rows[p] (the caller sums them over the index lists).
 * ---------------------------------------------------------------------------------------------- */
#define TEMP_SCORE_DISTMULT 0
#define TEMP_SCORE_COMPLEX 1
int temp_bilinear_query_fwd(int P, int d, int kind, const float* ent_rows, const int32_t* known_idx, const float* rel, const int32_t* rel_idx,
                            const int32_t* is_tail, float* q, void* stream);
int temp_bilinear_query_bwd(int P, int d, int kind, const float* ent_rows, const int32_t* known_idx, const float* rel, const int32_t* rel_idx,
                            const int32_t* is_tail, const float* d_q, float* d_known_rows, float* d_rel_rows, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Snapshot store (build_interpolation_graphs / get_train_val_test_graph_at_t keep one DGL graph per timestamp,
 * utils/dataset.py:151-232,268-305; dgl.batch rebuilds the union every call, models/DynamicRGCN.py:92).  Here every
 * snapshot's sorted / chunked edge views stay resident on the device; a batch's TempGraph arrays are assembled from them
 * by ONE launch: descriptor j copies member array src[0..len) to out[dst_off ..) shifted by `add`
 *   mode 0: v + add     mode 1: v >= 0 ? v + add : v     mode 2: t = table[add + aux[i]]; t >= 0 ? t + v : t
 * and piece p = elements [piece_start[p], piece_start[p] + TEMP_ASSEMBLE_PIECE) of descriptor piece_desc[p].
 * All pointers are DEVICE pointers (descs included).
 * ---------------------------------------------------------------------------------------------- */
#define TEMP_ASSEMBLE_PIECE 4096
typedef struct TempCopyDesc { const int32_t* src; const int32_t* aux; int32_t dst_off; int32_t len; int32_t add; int32_t mode; } TempCopyDesc;
int temp_assemble_views(int n_pieces, const int32_t* piece_desc, const int32_t* piece_start, const TempCopyDesc* descs, const int32_t* table,
                        int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device-side edge subsample of a resident snapshot: the training-time 50 % (target) / 80 %
 * (--random-dropout history) random edge subset with recomputed norms of DynamicRGCN.get_batch_graph_embeds
 * (models/DynamicRGCN.py:76-90; comp_deg_norm utils/utils.py:74-79), derived from the snapshot's resident sorted / chunked
 * views without a host rebuild or an upload.
 *   parent / child : packed view buffers of one snapshot with the layout of Snapshot.device_views (temp_amd/snapshot.py);
 *                    `child` must start as a copy of `parent`.  The call rewrites, in `child`, the a / b arrays of the three
 *                    views (kept edges moved to the front of every chunk, order preserved), their chunk_end arrays,
 *                    in_deg, out_deg and nnorm (1 / in_deg, 0 for isolated nodes); chunk tables, partial slots and fix-up
 *                    lists stay valid, so every temp_rgcn_* entry point runs on the child unchanged.
 *   eid [3][E]     : original edge id of every position of the by-dst / by-src / by-rel view.
 *   keep           : size of the subset; it is the set of the `keep` smallest values of a 64-bit counter hash of
 *                    (seed, edge id) -- a uniformly random `keep`-subset that depends on the seed only.
 *   keep_mask      : nullable [E] uint8 out, 1 = edge kept (original edge order).
 *   scratch        : >= 8 bytes of device memory per job.
 * `jobs` is a HOST array; all jobs of a batch run in three launches.
 * ---------------------------------------------------------------------------------------------- */
typedef struct TempSubsampleJob {
  int32_t n_nodes, n_edges, keep;
  uint64_t seed;
  const int32_t* parent; int32_t* child; const int32_t* eid;
  int32_t off_a[3], off_b[3], off_chunk_beg[3], off_chunk_end[3], off_chunk_seg[3], n_chunks[3];   /* word offsets inside a pack */
  int32_t off_in_deg, off_out_deg, off_nnorm;
  uint8_t* keep_mask; void* scratch;
} TempSubsampleJob;
int temp_subsample_views(int n_jobs, const TempSubsampleJob* jobs, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Filtered negative sampling (CorruptTriples.negative_sampling / corrupt_triple, utils/CorrptTriples.py:36-85):
 *   cand[row, 0] = truth[row];   cand[row, 1..K] = uniform draws over [0, N) that are not in the row's
 *   known-true set ids[lo[row] .. hi[row])  (global entity ids, ascending within a row; lo/hi NULL = no filter).
 * Exactly uniform over the entities outside the set: short sets by drawing the u-th entity of the complement directly,
 * long sets by rejection (binary search per attempt) with that as the fallback -- no unbounded resampling loop.
 * The draws are a counter-based hash of (seed, row, column, attempt): same seed => same samples, any launch shape.
 * cand: [R, 1+K] int32, fully written.
 * ---------------------------------------------------------------------------------------------- */
int temp_corrupt_sample(int R, int K, int N, uint64_t seed, const int32_t* truth, const int32_t* lo, const int32_t* hi, const int32_t* ids,
                        int32_t* cand, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Filtered ranking (EvaluationFilter.calc_metrics_single_graph / perturb_and_get_rank / sort_and_rank,
 * utils/evaluation.py:40-106).  scores [P, ld] = the P test triples scored against ALL N entities
 * (temp_linear with the folded query, trans_b = 1).  The reference overwrites the scores of the other
 * entities known to be true for the same (relation, known entity) with -10e6, applies a sigmoid and takes
 * the index of the target in a descending torch.sort.  Here:
 *   ranks[p] = 1 + #{ j : v_j > v_t  or  (v_j == v_t and j < target[p]) },
 *   v_j = sigmoid(scores[p,j]),  or 0 for j in the row's filter list  filt_ids[filt_ptr[p] .. filt_ptr[p+1])
 * (unique global ids; an entry equal to target[p] is ignored) -- the position in a STABLE descending order,
 * so exact sigmoid ties resolve by entity id where the reference's unstable sort leaves them arbitrary.
 * filt_ptr may be NULL (raw ranking); filt_ids may be NULL only when every list is empty.  ld % 4 == 0, ld >= N.  Integer output, deterministic.
 * ---------------------------------------------------------------------------------------------- */
int temp_filtered_rank(int P, int N, int ld, const float* scores, const int32_t* target, const int32_t* filt_ptr,
                       const int32_t* filt_ids, int32_t* ranks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * History attention of the self-attention encoder (SARGCNLayer.calc_result + attention,
 * models/SARGCN.py:25-53; callers models/SARGCN.py:39-62, models/SelfAttentionRGCN.py:88-96).
 * The reference builds a dense (n, T, D) tensor [history ..., current] (zero rows + a -10e9 additive
 * mask where a node was inactive) and projects all of it; here K / V are projected once per distinct
 * (snapshot, node) row into a table and every query row lists its active history rows:
 *   idx[i, t] (t < T-1) = row of the history table or -1 (masked: softmax weight exactly 0)
 *   position T-1 is the row's own current K / V (kc, vc).
 *   s[i,h,t] = <q[i,h,:], K[t][h,:]> / sqrt(d_k) + decay[t];  p = softmax_t;  o[i,h,:] = sum_t p V[t][h,:]
 *   out[i, d * heads + h] = o[i,h,d]      (the reference's transpose after squeeze, SARGCN.py:37)
 * heads must be 8 (SARGCN.py:21), D % 8 == 0, D <= 512, T <= 64.
 * fwd saves score [n, heads, T] (raw s, -inf where masked) and lse [n, heads] for bwd.
 * bwd: d_q / d_kc / d_vc fully written; d_kh / d_vh either ACCUMULATED with atomics (zero them first; rows are
 * shared between query rows) or, given the inverse maps, written once per table row; d_decay [T] accumulated when non-NULL.
 * ---------------------------------------------------------------------------------------------- */
typedef struct TempAttn {
  int32_t n, D, heads, T;
  const float* q;  int32_t ldq;                     /* [n, D] query projections */
  const float* kh; const float* vh; int32_t ldh;    /* history table K / V rows */
  const float* kc; const float* vc; int32_t ldc;    /* current-position K / V, row i */
  const int32_t* idx;                               /* [n, T-1] */
  const float* decay;                               /* [T] additive score bias or NULL */
} TempAttn;
int temp_sa_attn_fwd(const TempAttn* p, float* out, float* score, float* lse, void* stream);
int temp_sa_attn_bwd(const TempAttn* p, const float* out, const float* score, const float* lse, const float* d_out,
                     float* d_q, int ld_dq, float* d_kh, float* d_vh, int ld_dh, float* d_kc, float* d_vc, int ld_dc,
                     float* d_decay, int n_table, const int32_t* inv_ptr, const int32_t* inv_ref, float* ds_ws, void* stream);
/* Deterministic form: inv_ptr [n_table+1] / inv_ref = idx grouped by history-table row (ref = i * (T-1) + t, the static
 * maps of a prepared batch) and ds_ws [n * heads * T] scratch.  Then d_kh / d_vh rows [0, n_table) are each written once
 * by a second pass over the table (no atomics, no zero-fill needed).  With inv_ptr = NULL the atomic form above is used. */

/* Device-memory bandwidth probe used by bench.py to calibrate the achievable HBM peak
 * (float4 copy of `bytes` bytes, dst and src must not overlap). */
int temp_copy_probe(const void* src, void* dst, size_t bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-kernel timing for bench.py's roofline (not used by the product path).  Between begin and end
 * every kernel launch of this library is bracketed by HIP events on its launch stream.
 * temp_trace_end() synchronises the device, returns up to `capacity` (kernel id, milliseconds)
 * records in launch order and releases the events.  One trace at a time.
 * ---------------------------------------------------------------------------------------------- */
int temp_trace_begin(int capacity);
int temp_trace_end(int* kernel_ids /*host*/, float* ms /*host*/, int capacity, int* n_out /*host*/);
const char* temp_trace_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* TEMP_AMD_H */
