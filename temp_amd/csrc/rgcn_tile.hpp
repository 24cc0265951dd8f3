// LDS-tiled edge kernels for batched graphs of SMALL member snapshots (GDELT-shaped: hundreds of nodes, thousands of edges per
// snapshot; models/RGCN.py:91-104 and its autograd).
//
// The block-diagonal relation weights make the feature columns of a layer independent in groups of `so` (<= 4) columns, and no
// edge of a batched graph crosses its member snapshots.  So a workgroup takes ONE (member snapshot, slice of <= 16 float4
// columns): it stages the member's rows of that slice (500 nodes x 13 float4 = 104 KB), the slice of the relation weight table
// (17 KB), the member's edge ids as 16-/8-bit local indices (22 KB) and its chunk list, sorted by length (5 KB) in the CU's 160 KB
// of LDS, and then every gather of the aggregation is an LDS read instead of an L2 round trip (the L2-gather kernels above moved
// ~9 TB/s through L2 at a third of its peak and were bound by that latency).
//
// Lane layout: a wave holds four "walkers" of 16 lanes (the four lane groups in which the LDS serves a 16-byte read, see
// tile_lane); a walker owns one chunk (<= 64 edges of one destination, or <= 128 of one relation) and walks its edges in order, one
// float4 column per lane: exactly the per-chunk accumulation order of k_rgcn_agg_s / k_rgcn_dw_s, so results are BIT-IDENTICAL
// to those kernels (same chunk partials, same fix-up pass).  Chunk lengths are Zipf-distributed (a hub's chunks are full, most
// chunks hold a handful of edges): the block counting-sorts its chunks by length (LDS integer atomics: which chunk lands where
// inside a bucket varies, no result depends on it) and waves pull groups of four neighbouring chunks from a block-local queue, so
// the walkers of a wave finish together and waves balance dynamically.
//
// Per-edge scalars disappear: the d/dh kernel's nnorm[dst]^2 and the weight-gradient kernel's nnorm[dst]^2 are folded into the
// staged rows (the same single fp32 product the per-edge form computes).
#pragma once
#include "common.hpp"
#include <type_traits>

namespace temp {

#define TILE_LMAX 128                                        // longest chunk (TEMP_CHUNK_REL)
#define TILE_MISC_INTS (2 * (TILE_LMAX + 2) + 2)             // histogram, cursors, queue head
#define TILE_LDS_MAX (160 * 1024)                            // LDS of a CU: one workgroup takes what it needs of it
#define TILE_THREADS 1024

struct TileArgs {
  const int32_t* node_off;                                   // member tables (device), chunk_off of THIS view
  const int32_t* edge_off;
  const int32_t* chunk_off;
  const int32_t* fix_off;                                    // [n_members + 1] of THIS view, or nullptr: fix-up by its own launch
  int n_members;
  int fs4, n_slices;                                         // float4 columns of the widest slice (= LDS row stride), slices per row
  int off_x, off_g, off_w, off_ea, off_eb, off_cm, off_misc; // LDS byte offsets
  int lds_bytes;
  long long* prof;                                           // development only (temp_set_debug_buffer): 8 cycle stamps per block
};
#define TILE_STAMP(k) do { if (t.prof && threadIdx.x == 0) t.prof[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)

// LDS plan for a launch: the fewest slices (widest rows, <= 16 float4) whose staging fits one CU's LDS.
// rows2: the weight-gradient kernel stages two row sets (x and dz); w_rows: rows of the relation table (0: none staged).
inline bool tile_plan(const TempMembers& mb, int view, int D, int S, int rows2, int w_rows, int b_bytes, TileArgs* t) {
  if (mb.n_members <= 0 || !mb.node_off || !mb.edge_off || !mb.chunk_off) return false;
  if (mb.max_nodes <= 0 || mb.max_nodes > 65535 || mb.max_edges > 65535 || mb.max_chunks[view] > 65535) return false;
  const int D4 = D >> 2;
  for (int ns = ceil_div(D4, 16); ns <= D4; ++ns) {
    const int fs4 = ceil_div(D4, ns);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += align_up(bytes, 16); return (int)at; };
    t->off_x = take(((size_t)mb.max_nodes + (b_bytes == 0 ? 1 : 0)) * fs4 * 16);   // (+ the zero row of the packed walk)
    t->off_g = rows2 ? take((size_t)mb.max_nodes * fs4 * 16) : 0;
    t->off_w = w_rows ? take((size_t)w_rows * S * fs4 * 16) : 0;
    if (b_bytes == 0) {                                       // aggregation / d-dh: ONE packed word per edge (tile_stage_edges_packed)
      if (((size_t)mb.max_nodes + 1) * fs4 * 16 > 131071 || (size_t)w_rows * S * fs4 > 32767) continue;   // 17 + 15 bits of the word
      t->off_ea = take((size_t)mb.max_edges * 4 + 64);        // (+ a group of padding words: the walk reads one group ahead)
      t->off_eb = t->off_ea;
    } else {
      t->off_ea = take((size_t)mb.max_edges * 2 + 16);
      t->off_eb = take((size_t)mb.max_edges * b_bytes + 16);
    }
    t->off_cm = take((size_t)mb.max_chunks[view] * 8);
    t->off_misc = take((size_t)TILE_MISC_INTS * 4);
    if (off > TILE_LDS_MAX) continue;
    t->lds_bytes = (int)off;
    t->fs4 = fs4;
    t->n_slices = ns;
    t->n_members = mb.n_members;
    t->node_off = mb.node_off;
    t->edge_off = mb.edge_off;
    t->chunk_off = mb.chunk_off + (size_t)view * (mb.n_members + 1);
    // in-block fix-up: node views only, and only while no entry can be "long" for k_fixup<4, 4, 256> (its block-cooperative
    // walk sums in another order; a member's hub has at most max_edges / TEMP_CHUNK partial rows)
    t->fix_off = (view < 2 && mb.fix_off && mb.max_edges / TEMP_CHUNK < 256) ? mb.fix_off + (size_t)view * (mb.n_members + 1) : nullptr;
    t->prof = nullptr;
    return true;
  }
  return false;
}

struct TileBlock {
  int m, slice, n0, nm, e0, em, c0, nc, f4_0, nf4;
};

__device__ __forceinline__ bool tile_block(const TileArgs& t, int D4, TileBlock& b) {
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;   // the slices of one member run on one XCD (they share its rows' lines)
  b.m = xcd + 8 * (local / t.n_slices);
  b.slice = local % t.n_slices;
  if (b.m >= t.n_members) return false;
  b.n0 = t.node_off[b.m]; b.nm = t.node_off[b.m + 1] - b.n0;
  b.e0 = t.edge_off[b.m]; b.em = t.edge_off[b.m + 1] - b.e0;
  b.c0 = t.chunk_off[b.m]; b.nc = t.chunk_off[b.m + 1] - b.c0;
  // an EMPTY member (a timestamp with no train facts: no nodes, or nodes without edges) has nothing to aggregate -- rows without
  // in-edges are never read from the aggregation output -- and its clamped staging indices would point before / past the views
  if (b.nm <= 0 || b.nc <= 0 || b.em <= 0) return false;
  b.f4_0 = (b.slice * D4) / t.n_slices;                      // even slices: widths differ by at most one float4
  b.nf4 = ((b.slice + 1) * D4) / t.n_slices - b.f4_0;
  return true;
}

// A wave's four WALKERS are the four 16-lane groups in which the LDS serves a ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32: one LDS cycle per group when its lanes hit distinct banks).  The lanes of a walker read one
// CONTIGUOUS run of <= 256 bytes (its <= 16 float4 of one staged row): every bank at most once, whatever the row -- so the random
// row gathers of the aggregation run at the full 256 B/clk of the LDS.  (With walkers of consecutive lanes several rows share a
// group and their bank ranges overlap at random: measured 2.3x slower.)  -> walker 0..3, column 0..15 of the lane.
__device__ __forceinline__ void tile_lane(int lane, int& g, int& lr) {
  const int l = lane & 31;
  int grp, idx;
  if (l < 4) { grp = 0; idx = l; }
  else if (l < 12) { grp = 1; idx = l - 4; }
  else if (l < 16) { grp = 0; idx = l - 8; }
  else if (l < 20) { grp = 1; idx = l - 8; }
  else if (l < 28) { grp = 0; idx = l - 12; }
  else { grp = 1; idx = l - 16; }
  g = (lane >> 5) * 2 + grp;
  lr = idx;
}

// rows of one slice -> LDS (row i of the member at Xs[i * fs4 + lr]); `ids` (nullable) maps node -> table row; SCALE: x nnorm[node]^2
template <bool SCALE>
__device__ __forceinline__ void tile_stage_rows(float4* Xs, const TileArgs& t, const TileBlock& b, const float* __restrict__ src, int ld,
                                                const int32_t* __restrict__ ids, const float* __restrict__ nnorm) {
  const int w = b.nf4, rstep = blockDim.x / w;                // consecutive threads read consecutive float4 of a row
  const int r0 = threadIdx.x / w, lr = threadIdx.x - r0 * w;
  if (r0 >= rstep) return;
  const float* base = src + (size_t)(b.f4_0 + lr) * 4;
  const int fs4 = t.fs4;
  constexpr int U = 8;                                       // rows in flight per thread (every batch predicated: a member has few passes)
  for (int i = r0; i < b.nm; i += U * rstep) {
    int row[U];
    float4 x[U];
    float nn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int node = b.n0 + min(i + u * rstep, b.nm - 1);  // past the end: the last row again (not stored)
      row[u] = ids ? ids[node] : node;
      nn[u] = SCALE ? nnorm[node] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = ld4(base + (size_t)row[u] * ld);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * rstep < b.nm) Xs[(i + u * rstep) * fs4 + lr] = SCALE ? scale4(x[u], nn[u] * nn[u]) : x[u];
  }
}

// the member's edge ids of a view as local 16-bit (a) and BT-sized (b) indices; a_sub / b_sub: what to subtract (node offset or 0)
template <class BT>
__device__ __forceinline__ void tile_stage_edges(unsigned short* Ea, BT* Eb, const TempEdgeView& v, const TileBlock& b, int a_sub, int b_sub) {
  const int nthr = blockDim.x;
  constexpr int U = 8;
  for (int i = threadIdx.x; i < b.em; i += U * nthr) {
    int a[U], bb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = b.e0 + min(i + u * nthr, b.em - 1);
      a[u] = v.a[e];
      bb[u] = v.b[e];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * nthr < b.em) { Ea[i + u * nthr] = (unsigned short)(a[u] - a_sub); Eb[i + u * nthr] = (BT)(bb[u] - b_sub); }
  }
}

// The aggregation / d-dh walk keeps ONE word per edge holding both LDS offsets, ready to use --
// bits 0-16: BYTE offset of the edge's staged row  (a_local * fs4 * 16), bits 17-31: weight offset  b * S * fs4  in units of 16
// bytes (Ws + that = the relation's block rows).  The walk was bound by instruction ISSUE, not by the LDS (4 waves per SIMD x ~22
// instructions per edge step: 47 k cycles per block against 26 k of LDS reads): two id reads, two clamps, two 24-bit multiplies
// and their address adds per edge became one read, a mask and a shift (k_rgcn_agg_t builds the words while staging).

// Chunk list of the member, counting-sorted by length (longest first) into cm[]:
//   .x = first edge (relative to the member's edge range) | segment (relative to seg_sub) << 16      .y = partial slot (0xffffff: none) | length << 24
// misc: [0, LMAX+2) histogram, [LMAX+2, 2 LMAX+4) cursors, then the queue head.  Must be zero on entry (and the zeroing visible:
// a barrier before the call); ends with a barrier.  The chunk arrays are read once (a thread keeps up to two chunks in registers).
// Workgroup barrier that waits for this wave's LDS operations only: __syncthreads() also waits for every outstanding GLOBAL load
// (s_waitcnt vmcnt(0)), which would end the overlap of the staging loads with the chunk sort.  Only where the data handed over
// lives in LDS.
__device__ __forceinline__ void tile_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The same sort with the thread's FIRST chunk record already in registers (loaded by the caller ahead of its other staging loads;
// `crec` = {first edge, end, segment, slot}, valid when tid < b.nc) and LDS-only barriers: no wait on the caller's loads in flight.
__device__ __forceinline__ void tile_sort_chunks_pre(uint2* cm, int* misc, const TempEdgeView& v, const TileBlock& b, int seg_sub, const int (&crec)[4]) {
  const int nthr = blockDim.x, tid = threadIdx.x;
  int* hist = misc;
  int* cur = misc + TILE_LMAX + 2;
  const bool mine = tid < b.nc;
  const int len0 = min(max(crec[1] - crec[0], 0), TILE_LMAX);
  if (mine) atomicAdd(&hist[TILE_LMAX - len0], 1);
  for (int i = tid + nthr; i < b.nc; i += nthr) {             // (members with more than 1024 chunks)
    const int beg = v.chunk_beg[b.c0 + i], len = min(max(v.chunk_end[b.c0 + i] - beg, 0), TILE_LMAX);
    atomicAdd(&hist[TILE_LMAX - len], 1);
  }
  tile_barrier_lds();
  if (tid <= TILE_LMAX) {
    int s = 0;
    for (int q = 0; q < tid; ++q) s += hist[q];
    cur[tid] = s;
  }
  tile_barrier_lds();
  if (mine)
    cm[atomicAdd(&cur[TILE_LMAX - len0], 1)] = make_uint2((unsigned)(crec[0] - b.e0) | ((unsigned)(crec[2] - seg_sub) << 16), ((unsigned)crec[3] & 0xffffffu) | ((unsigned)len0 << 24));
  for (int i = tid + nthr; i < b.nc; i += nthr) {
    const int beg = v.chunk_beg[b.c0 + i], len = min(max(v.chunk_end[b.c0 + i] - beg, 0), TILE_LMAX);
    const int seg = v.chunk_seg[b.c0 + i], slot = v.chunk_slot[b.c0 + i];
    cm[atomicAdd(&cur[TILE_LMAX - len], 1)] = make_uint2((unsigned)(beg - b.e0) | ((unsigned)(seg - seg_sub) << 16), ((unsigned)slot & 0xffffffu) | ((unsigned)len << 24));
  }
  tile_barrier_lds();
}

__device__ __forceinline__ void tile_sort_chunks(uint2* cm, int* misc, const TempEdgeView& v, const TileBlock& b, int seg_sub) {
  const int nthr = blockDim.x, tid = threadIdx.x;
  int* hist = misc;
  int* cur = misc + TILE_LMAX + 2;
  uint2 keep[2];
  int klen[2] = {-1, -1};
  for (int i = tid, u = 0; i < b.nc; i += nthr, ++u) {
    const int beg = v.chunk_beg[b.c0 + i], len = min(max(v.chunk_end[b.c0 + i] - beg, 0), TILE_LMAX);
    atomicAdd(&hist[TILE_LMAX - len], 1);
    if (u < 2) {
      const int seg = v.chunk_seg[b.c0 + i], slot = v.chunk_slot[b.c0 + i];
      keep[u] = make_uint2((unsigned)(beg - b.e0) | ((unsigned)(seg - seg_sub) << 16), ((unsigned)slot & 0xffffffu) | ((unsigned)len << 24));
      klen[u] = len;
    }
  }
  __syncthreads();
  if (tid <= TILE_LMAX) {
    int s = 0;
    for (int q = 0; q < tid; ++q) s += hist[q];
    cur[tid] = s;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (klen[u] >= 0) cm[atomicAdd(&cur[TILE_LMAX - klen[u]], 1)] = keep[u];
  for (int i = tid + 2 * nthr; i < b.nc; i += nthr) {         // (members with more than 2 x 1024 chunks: re-read)
    const int beg = v.chunk_beg[b.c0 + i], len = min(max(v.chunk_end[b.c0 + i] - beg, 0), TILE_LMAX);
    const int seg = v.chunk_seg[b.c0 + i], slot = v.chunk_slot[b.c0 + i];
    cm[atomicAdd(&cur[TILE_LMAX - len], 1)] = make_uint2((unsigned)(beg - b.e0) | ((unsigned)(seg - seg_sub) << 16), ((unsigned)slot & 0xffffffu) | ((unsigned)len << 24));
  }
  __syncthreads();
}

// Forward aggregation (view = by-dst: a = src node, b = relation; result x nnorm[seg]^2) and d/dh (view = by-src: a = dst node,
// b = relation; rows pre-scaled by nnorm[dst]^2, transposed blocks).  BT: storage of the relation ids in LDS.
// VAR (development ablations, 0 in the product path): 1 no row reads, 2 no weight reads, 4 no edge-id reads, 8 walkers = consecutive lanes
template <int S, int MODE, class BT, int VAR = 0>
__global__ void __launch_bounds__(TILE_THREADS) k_rgcn_agg_t(TempEdgeView v, TileArgs t, const float* __restrict__ feat, int ldf,
                                                             const int32_t* __restrict__ feat_ids, const float* __restrict__ W, int n_rel_rows,
                                                             const float* __restrict__ nnorm, int D, float* __restrict__ out,
                                                             float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];
  float4* Xs = reinterpret_cast<float4*>(tile_lds + t.off_x);
  float4* Ws = reinterpret_cast<float4*>(tile_lds + t.off_w);
  unsigned* Ep = reinterpret_cast<unsigned*>(tile_lds + t.off_ea);
  uint2* cm = reinterpret_cast<uint2*>(tile_lds + t.off_cm);
  int* misc = reinterpret_cast<int*>(tile_lds + t.off_misc);
  const int D4 = D >> 2;
  TileBlock b;
  if (!tile_block(t, D4, b)) return;
  const int tid = threadIdx.x, nthr = blockDim.x, fs4 = t.fs4;
  TILE_STAMP(0);
  // ---- staging.  The FIRST batch of every stream -- eight rows, one weight item, eight edges per thread: all of a GDELT-sized
  // member but half of its edges -- is requested before anything waits, and the chunk sort (global loads of the chunk records,
  // LDS atomics, three barriers) runs while those loads are in flight: one memory round trip where the four phases used to pay
  // one each (rows 9.5 k + weights 3.2 k + edges 2.8 k + sort 4.9 k cycles of a 57 k-cycle block).
  for (int i = tid; i < TILE_MISC_INTS; i += nthr) misc[i] = 0;
  int crec[4];                                                // this thread's chunk record FIRST: the sort waits for it, and the
  {                                                           // load counter is in order -- what is requested behind it stays in flight
    const int ci = b.c0 + min(tid, max(b.nc - 1, 0));
    crec[0] = v.chunk_beg[ci]; crec[1] = v.chunk_end[ci]; crec[2] = v.chunk_seg[ci]; crec[3] = v.chunk_slot[ci];
  }
  constexpr int RU = 8;
  const int rw = b.nf4, rstep = nthr / rw;
  const int rr0 = min(tid / rw, rstep - 1), rlr = tid - (tid / rw) * rw;
  const bool rower = tid / rw < rstep;                        // (the last few threads repeat the last row group's loads and store nothing)
  const float* rbase = feat + (size_t)(b.f4_0 + rlr) * 4;
  float4 rx[RU];
  float rnn[RU];
  {
    int row[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int node = b.n0 + min(rr0 + u * rstep, b.nm - 1);  // past the end: the last row again (not stored)
      row[u] = feat_ids ? feat_ids[node] : node;
      rnn[u] = (MODE == MODE_DX) ? nnorm[node] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) rx[u] = ld4(rbase + (size_t)row[u] * ldf);
  }
  const float4* W4 = reinterpret_cast<const float4*>(W);
  const int w_total = n_rel_rows * S * fs4;
  auto w_src = [&](int q, bool& ok) {                          // Ws[(r * S + j) * fs4 + lr] = W4[r * D4 * S + (f4_0 + lr) * S + j]
    const int rj = q / fs4, lr = q - rj * fs4;
    const int r = rj / S, j = rj - r * S;
    ok = q < w_total && lr < b.nf4;
    return W4 + (ok ? (size_t)r * D4 * S + (size_t)(b.f4_0 + lr) * S + j : 0);
  };
  bool w_ok0;
  const float4 w0 = *w_src(tid, w_ok0);
  constexpr int EU = 8;
  int ea0[EU], eb0[EU];
#pragma unroll
  for (int u = 0; u < EU; ++u) {
    const int e = b.e0 + min(tid + u * nthr, max(b.em - 1, 0));
    ea0[u] = v.a[e];
    eb0[u] = v.b[e];
  }
  TILE_STAMP(1);
  tile_barrier_lds();                                         // misc zeroed
  TILE_STAMP(2);
  tile_sort_chunks_pre(cm, misc, v, b, b.n0, crec);
  TILE_STAMP(3);
  // ---- commit the first batches, then whatever a larger member has beyond them
  if (rower) {
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (rr0 + u * rstep < b.nm) Xs[(rr0 + u * rstep) * fs4 + rlr] = (MODE == MODE_DX) ? scale4(rx[u], rnn[u] * rnn[u]) : rx[u];
    for (int i = rr0 + RU * rstep; i < b.nm; i += rstep) {
      const int node = b.n0 + i;
      const int row = feat_ids ? feat_ids[node] : node;
      const float4 x = ld4(rbase + (size_t)row * ldf);
      if (MODE == MODE_DX) { const float nn = nnorm[node]; Xs[i * fs4 + rlr] = scale4(x, nn * nn); }
      else Xs[i * fs4 + rlr] = x;
    }
  }
  for (int i = tid; i < fs4; i += nthr) Xs[b.nm * fs4 + i] = zero4();          // the null edge's row
  if (w_ok0) Ws[tid] = w0;
  for (int q = tid + nthr; q < w_total; q += nthr) {
    bool ok;
    const float4* p = w_src(q, ok);
    if (ok) Ws[q] = *p;
  }
  {
    const unsigned row_b = (unsigned)fs4 << 4, wrow16 = (unsigned)(S * fs4);
#pragma unroll
    for (int u = 0; u < EU; ++u)
      if (tid + u * nthr < b.em) Ep[tid + u * nthr] = (unsigned)(ea0[u] - b.n0) * row_b | ((unsigned)eb0[u] * wrow16) << 17;
    for (int i = tid + EU * nthr; i < b.em; i += nthr) Ep[i] = (unsigned)(v.a[b.e0 + i] - b.n0) * row_b | ((unsigned)v.b[b.e0 + i] * wrow16) << 17;
    if (tid < 8) Ep[b.em + tid] = 0u;                         // the walk reads one group ahead
  }
  TILE_STAMP(4);
  __syncthreads();
  TILE_STAMP(5);

  const int lane = tid & 63;
  int g, lr;
  tile_lane(lane, g, lr);
  if (VAR & 8) { g = lane >> 4; lr = lane & 15; }
  const bool lane_ok = lr < b.nf4;
  int* queue = misc + 2 * (TILE_LMAX + 2);
  // byte addressing with 24-bit multiplies (full-rate v_mad_u32_u24; a 32-bit v_mul_lo_u32 issues at a quarter of that)
  const unsigned char* xl = reinterpret_cast<const unsigned char*>(Xs + lr);
  const unsigned char* wl = reinterpret_cast<const unsigned char*>(Ws + lr);
  const unsigned xrow = (unsigned)fs4 * 16u, wrow_b = (unsigned)(S * fs4) * 16u;
  for (;;) {
    int task = 0;
    if (lane == 0) task = atomicAdd(queue, 1);
    task = __builtin_amdgcn_readfirstlane(task);
    const int k = task * 4 + g;
    if (task * 4 >= b.nc) break;
    const bool has = lane_ok && k < b.nc;
    const uint2 mt = has ? cm[k] : make_uint2(0u, 0xffffffu);
    const int beg = mt.x & 0xffffu, segl = mt.x >> 16, len = mt.y >> 24;
    const unsigned slot = mt.y & 0xffffffu;
    float nn = 0.f;
    if (MODE == MODE_FWD && has) nn = nnorm[b.n0 + segl];      // in flight during the walk
    float4 acc = zero4();
    // Groups of four edges: the four packed id words of the NEXT group are requested (two ds_read2_b32: no clamp, the array is
    // padded) before this group's twelve row / weight reads and products; an edge's two LDS addresses are a mask / shift of its
    // word + the lane's base.  Edges are accumulated in order, one at a time: the same sums as the gather kernels.
    constexpr int U = 4;
    const unsigned* ep = Ep + beg;
    const unsigned null_word = (unsigned)b.nm * xrow;         // the zero row behind the member's rows, weights of relation 0: adds +0
    unsigned idw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idw[u] = ep[u];
    auto group = [&](int j, auto full_c) {                      // four edges; FULL: all four inside the chunk for every walker
      constexpr bool FULL = decltype(full_c)::value;
      float4 x[U], w[U][S];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // past the chunk's end: the null edge instead of a predicate on the products (one select here against four there)
        const unsigned word = (FULL || j + u < len) ? idw[u] : null_word;
        const unsigned xo = word & 0x1ffffu, wo = __builtin_amdgcn_ubfe(word, 17, 15) << 4;
        x[u] = (VAR & 1) ? make_float4((float)xo, 1.f, 2.f, 3.f) : *reinterpret_cast<const float4*>(xl + xo);
#pragma unroll
        for (int q = 0; q < S; ++q)
          w[u][q] = (VAR & 2) ? make_float4((float)wo, 1.f, 0.5f, (float)q) : *reinterpret_cast<const float4*>(wl + wo + q * xrow);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) idw[u] = ep[j + U + u];
#pragma unroll
      for (int u = 0; u < U; ++u) block_mac<S, MODE>(acc, x[u], w[u], 1.f);
    };
    for (int j = 0; j < len; j += U) {                        // (per-walker trip count: the chunks of a task are neighbours in the sort)
      // every active walker of the wave has a whole group left: no selects (wave-uniform branch)
      if (__builtin_amdgcn_ballot_w64(j + U > len) == 0) group(j, std::true_type());
      else group(j, std::false_type());
    }
    if (has) {
      if (MODE == MODE_FWD) acc = scale4(acc, nn * nn);
      float* dst = (slot == 0xffffffu) ? out + (size_t)(b.n0 + segl) * D : partial + (size_t)slot * D;
      st4(dst + (b.f4_0 + lr) * 4, acc);
    }
  }
  TILE_STAMP(6);
  // ---- the member's multi-chunk segments: all their partial rows were written by THIS block (a member's chunks never leave
  // it), so their ordered sums are taken here instead of by a k_fixup launch behind the kernel: same walk, same order, same bits
  if (t.fix_off) {
    __syncthreads();                                          // (workgroup-scope: the walkers' partial rows are visible to the block)
    const int f0 = t.fix_off[b.m], nfx = t.fix_off[b.m + 1] - f0;
    for (int i = tid; i < nfx * b.nf4; i += nthr) {
      const int fi = f0 + i / b.nf4, c = (b.f4_0 + i % b.nf4) * 4;
      const int seg = v.fix_seg[fi], s0 = v.fix_slot[fi], cnt = v.fix_cnt[fi];
      // fixup_walk's additions in fixup_walk's order, with 32 instead of 8 rows in flight: a hub's hundred partial rows are four
      // memory round trips on the block's tail instead of fifteen
      const float* pr = partial + (size_t)s0 * D + c;
      float4 acc = zero4();
      int r = 0;
      for (; r + 32 <= cnt; r += 32) {
        float4 q[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) q[u] = ld4(pr + (size_t)(r + u) * D);
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = add4(acc, q[u]);
      }
      for (; r + 8 <= cnt; r += 8) {
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = ld4(pr + (size_t)(r + u) * D);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = add4(acc, q[u]);
      }
      for (; r < cnt; ++r) acc = add4(acc, ld4(pr + (size_t)r * D));
      st4(out + (size_t)seg * D + c, acc);
    }
  }
  if (t.prof) { __syncthreads(); TILE_STAMP(7); }
}

// d/dweight (view = by-rel: a = src node, b = dst node, segment = relation row): x rows and dz rows (pre-scaled by nnorm[dst]^2)
// staged; a walker accumulates the S x S outer products of its blocks over one relation chunk.
template <int S>
__global__ void __launch_bounds__(TILE_THREADS) k_rgcn_dw_t(TempEdgeView v, TileArgs t, const float* __restrict__ x, const int32_t* __restrict__ x_ids,
                                                            const float* __restrict__ dz, const float* __restrict__ nnorm, int D,
                                                            float* __restrict__ dW, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];
  float4* Xs = reinterpret_cast<float4*>(tile_lds + t.off_x);
  float4* Gs = reinterpret_cast<float4*>(tile_lds + t.off_g);
  unsigned short* Ea = reinterpret_cast<unsigned short*>(tile_lds + t.off_ea);
  unsigned short* Eb = reinterpret_cast<unsigned short*>(tile_lds + t.off_eb);
  uint2* cm = reinterpret_cast<uint2*>(tile_lds + t.off_cm);
  int* misc = reinterpret_cast<int*>(tile_lds + t.off_misc);
  const int D4 = D >> 2;
  TileBlock b;
  if (!tile_block(t, D4, b)) return;
  const int tid = threadIdx.x, nthr = blockDim.x, fs4 = t.fs4;
  for (int i = tid; i < TILE_MISC_INTS; i += nthr) misc[i] = 0;
  tile_stage_rows<false>(Xs, t, b, x, D, x_ids, nnorm);
  tile_stage_rows<true>(Gs, t, b, dz, D, nullptr, nnorm);
  tile_stage_edges<unsigned short>(Ea, Eb, v, b, b.n0, b.n0);
  __syncthreads();
  tile_sort_chunks(cm, misc, v, b, 0);

  const int lane = tid & 63;
  int g, lr;
  tile_lane(lane, g, lr);
  const bool lane_ok = lr < b.nf4;
  int* queue = misc + 2 * (TILE_LMAX + 2);
  const unsigned char* xl = reinterpret_cast<const unsigned char*>(Xs + lr);
  const unsigned char* gl = reinterpret_cast<const unsigned char*>(Gs + lr);
  const unsigned xrow = (unsigned)fs4 * 16u;
  const int wrow = D * S;
  for (;;) {
    int task = 0;
    if (lane == 0) task = atomicAdd(queue, 1);
    task = __builtin_amdgcn_readfirstlane(task);
    const int k = task * 4 + g;
    if (task * 4 >= b.nc) break;
    const bool has = lane_ok && k < b.nc;
    const uint2 mt = has ? cm[k] : make_uint2(0u, 0xffffffu);
    const int beg = mt.x & 0xffffu, seg = mt.x >> 16, len = mt.y >> 24;
    const unsigned slot = mt.y & 0xffffffu;
    float4 acc[S];
#pragma unroll
    for (int q = 0; q < S; ++q) acc[q] = zero4();
    const unsigned short* ea = Ea + beg;
    const unsigned short* eb = Eb + beg;
    for (int j = 0; j < len; ++j) {
      const float4 xx = *reinterpret_cast<const float4*>(xl + __umul24((unsigned)ea[j], xrow));
      const float4 gg = *reinterpret_cast<const float4*>(gl + __umul24((unsigned)eb[j], xrow));
      if (S == 1) {
        acc[0].x = fmaf(xx.x, gg.x, acc[0].x);
        acc[0].y = fmaf(xx.y, gg.y, acc[0].y);
        acc[0].z = fmaf(xx.z, gg.z, acc[0].z);
        acc[0].w = fmaf(xx.w, gg.w, acc[0].w);
      } else if (S == 2) {
        acc[0].x = fmaf(xx.x, gg.x, acc[0].x);
        acc[0].y = fmaf(xx.x, gg.y, acc[0].y);
        acc[0].z = fmaf(xx.y, gg.x, acc[0].z);
        acc[0].w = fmaf(xx.y, gg.y, acc[0].w);
        acc[S > 1 ? 1 : 0].x = fmaf(xx.z, gg.z, acc[S > 1 ? 1 : 0].x);
        acc[S > 1 ? 1 : 0].y = fmaf(xx.z, gg.w, acc[S > 1 ? 1 : 0].y);
        acc[S > 1 ? 1 : 0].z = fmaf(xx.w, gg.z, acc[S > 1 ? 1 : 0].z);
        acc[S > 1 ? 1 : 0].w = fmaf(xx.w, gg.w, acc[S > 1 ? 1 : 0].w);
      } else {
        acc[0] = fma4(xx.x, gg, acc[0]);
        acc[S > 1 ? 1 : 0] = fma4(xx.y, gg, acc[S > 1 ? 1 : 0]);
        acc[S > 2 ? 2 : 0] = fma4(xx.z, gg, acc[S > 2 ? 2 : 0]);
        acc[S > 3 ? 3 : 0] = fma4(xx.w, gg, acc[S > 3 ? 3 : 0]);
      }
    }
    if (has) {
      float* dst = ((slot == 0xffffffu) ? dW + (size_t)seg * wrow : partial + (size_t)slot * wrow) + (size_t)(b.f4_0 + lr) * 4 * S;
#pragma unroll
      for (int q = 0; q < S; ++q) st4(dst + 4 * q, acc[q]);
    }
  }
}

// d/dweight with ONE staged row set: the dz rows (pre-scaled by nnorm[dst]^2) of the slice in LDS -- 13 float4 per row, four
// slices, as in the d/dh kernel -- and the x row of every edge read from global memory (L2) through the edge's RESOLVED row index
// (node -> table row done while staging: layer 1 reads the embedding table).  Half the L2 gathers of k_rgcn_dw_s (which fetches
// both rows per edge) and none of the narrow slices of k_rgcn_dw_t (two row sets in LDS: 7-8 float4 wide).  U edges of a walker
// are in flight; the products are added in edge order, so chunk partials are BIT-IDENTICAL to k_rgcn_dw_s / k_rgcn_dw_t.
template <int S>
__global__ void __launch_bounds__(TILE_THREADS) k_rgcn_dw_h(TempEdgeView v, TileArgs t, const float* __restrict__ x, const int32_t* __restrict__ x_ids,
                                                            const float* __restrict__ dz, const float* __restrict__ nnorm, int D,
                                                            float* __restrict__ dW, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];
  float4* Gs = reinterpret_cast<float4*>(tile_lds + t.off_x);
  unsigned short* Eg = reinterpret_cast<unsigned short*>(tile_lds + t.off_ea);      // local dst node of the edge (row of Gs)
  unsigned* Ex = reinterpret_cast<unsigned*>(tile_lds + t.off_eb);                  // global x row of the edge
  uint2* cm = reinterpret_cast<uint2*>(tile_lds + t.off_cm);
  int* misc = reinterpret_cast<int*>(tile_lds + t.off_misc);
  const int D4 = D >> 2;
  TileBlock b;
  if (!tile_block(t, D4, b)) return;
  const int tid = threadIdx.x, nthr = blockDim.x, fs4 = t.fs4;
  for (int i = tid; i < TILE_MISC_INTS; i += nthr) misc[i] = 0;
  tile_stage_rows<true>(Gs, t, b, dz, D, nullptr, nnorm);
  {
    constexpr int U = 8;
    for (int i = tid; i < b.em; i += U * nthr) {
      int a[U], bb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = b.e0 + min(i + u * nthr, b.em - 1);
        a[u] = v.a[e];
        bb[u] = v.b[e];
      }
      if (x_ids) {
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = x_ids[a[u]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i + u * nthr < b.em) { Ex[i + u * nthr] = (unsigned)a[u]; Eg[i + u * nthr] = (unsigned short)(bb[u] - b.n0); }
    }
  }
  __syncthreads();
  tile_sort_chunks(cm, misc, v, b, 0);

  const int lane = tid & 63;
  int g, lr;
  tile_lane(lane, g, lr);
  const bool lane_ok = lr < b.nf4;
  int* queue = misc + 2 * (TILE_LMAX + 2);
  const unsigned char* gl = reinterpret_cast<const unsigned char*>(Gs + lr);
  const float* xl = x + (size_t)(b.f4_0 + (lane_ok ? lr : 0)) * 4;
  const unsigned xrow = (unsigned)fs4 * 16u;
  const int wrow = D * S;
  constexpr int U = 8;
  for (;;) {
    int task = 0;
    if (lane == 0) task = atomicAdd(queue, 1);
    task = __builtin_amdgcn_readfirstlane(task);
    const int k = task * 4 + g;
    if (task * 4 >= b.nc) break;
    const bool has = lane_ok && k < b.nc;
    const uint2 mt = has ? cm[k] : make_uint2(0u, 0xffffffu);
    const int beg = mt.x & 0xffffu, seg = mt.x >> 16, len = mt.y >> 24;
    const unsigned slot = mt.y & 0xffffffu;
    float4 acc[S];
#pragma unroll
    for (int q = 0; q < S; ++q) acc[q] = zero4();
    const unsigned short* eg = Eg + beg;
    const unsigned* ex = Ex + beg;
    for (int j = 0; j < len; j += U) {
      float4 xx[U], gg[U];
      unsigned xr[U], gr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = min(j + u, len - 1);                   // past the end: the last edge again (its product is not added)
        xr[u] = ex[jj];
        gr[u] = eg[jj];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) xx[u] = ld4(xl + (size_t)xr[u] * D);
#pragma unroll
      for (int u = 0; u < U; ++u) gg[u] = *reinterpret_cast<const float4*>(gl + __umul24(gr[u], xrow));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j + u < len) {
          const float4 a4 = xx[u], g4 = gg[u];
          if (S == 1) {
            acc[0].x = fmaf(a4.x, g4.x, acc[0].x);
            acc[0].y = fmaf(a4.y, g4.y, acc[0].y);
            acc[0].z = fmaf(a4.z, g4.z, acc[0].z);
            acc[0].w = fmaf(a4.w, g4.w, acc[0].w);
          } else if (S == 2) {
            acc[0].x = fmaf(a4.x, g4.x, acc[0].x);
            acc[0].y = fmaf(a4.x, g4.y, acc[0].y);
            acc[0].z = fmaf(a4.y, g4.x, acc[0].z);
            acc[0].w = fmaf(a4.y, g4.y, acc[0].w);
            acc[S > 1 ? 1 : 0].x = fmaf(a4.z, g4.z, acc[S > 1 ? 1 : 0].x);
            acc[S > 1 ? 1 : 0].y = fmaf(a4.z, g4.w, acc[S > 1 ? 1 : 0].y);
            acc[S > 1 ? 1 : 0].z = fmaf(a4.w, g4.z, acc[S > 1 ? 1 : 0].z);
            acc[S > 1 ? 1 : 0].w = fmaf(a4.w, g4.w, acc[S > 1 ? 1 : 0].w);
          } else {
            acc[0] = fma4(a4.x, g4, acc[0]);
            acc[S > 1 ? 1 : 0] = fma4(a4.y, g4, acc[S > 1 ? 1 : 0]);
            acc[S > 2 ? 2 : 0] = fma4(a4.z, g4, acc[S > 2 ? 2 : 0]);
            acc[S > 3 ? 3 : 0] = fma4(a4.w, g4, acc[S > 3 ? 3 : 0]);
          }
        }
      }
    }
    if (has) {
      float* dst = ((slot == 0xffffffu) ? dW + (size_t)seg * wrow : partial + (size_t)slot * wrow) + (size_t)(b.f4_0 + lr) * 4 * S;
#pragma unroll
      for (int q = 0; q < S; ++q) st4(dst + 4 * q, acc[q]);
    }
  }
}

}  // namespace temp
