// Feature-sliced edge kernels for LARGE graphs whose relation table does not fit LDS (the HBM regime the BASELINE metric is named
// after: S-hbm = 2^20 nodes, 2^24 edges, 2 x 230 relation rows of 1.6 KB = 736 KB; models/RGCN.py:91-104 and its autograd).
//
// With the table in global memory every edge reads its 1.6 KB of block weights through L2 next to the 0.8 KB source row it
// gathers: the forward ran at 4.2 TB/s algorithmic against 8.5 TB/s on the SAME graph with 20 relations (table in LDS), i.e. the
// weight reads, not HBM, bound it.  The block-diagonal weights make the feature columns independent in groups of `so` <= 4, so a
// workgroup takes ONE slice of <= 16 float4 columns for a range of chunks: the slice of the whole table (460 rows x 2 x 10
// float4 = 147 KB at D = 200: five slices) stays in LDS for the life of the (persistent) block, a gather fetches the slice's
// 160 bytes of the source row, and the ids of an edge are read once per slice (40 of 800 bytes).  The slices of one chunk range
// run on ONE XCD at the same time, so the 64-byte blocks two neighbouring slices share at their seam are L2 hits.
//
// Lane layout as in rgcn_tile.hpp: a wave holds four walkers (the four 16-lane groups in which the LDS serves a 16-byte read:
// conflict-free weight reads), a walker owns one chunk and walks its edges IN ORDER, one float4 column per lane -- the per-chunk
// accumulation order of k_rgcn_agg_s / k_rgcn_agg<lpr = 64>, so results are BIT-IDENTICAL to those kernels (same chunk partials,
// same fix-up pass).  The ids (and, for d/dh, nnorm[dst]^2) of sixteen edges are loaded by the walker's sixteen lanes at once and
// handed round with ds_bpermute; eight row gathers are in flight per walker.
//
// STATUS (round 3): correct (bit-identical, tests/test_gpu_parity_r2.py::test_sliced_edge_kernels_bit_identical_gpu) but SLOWER
// than the kernels that read the table through L2 -- forward 3.03 against 2.75 ms on S-hbm -- and therefore opt-in
// (TEMP_OPT_RGCN_SLICE).  Every chunk is visited once per slice, and a visit is a chain of dependent loads (chunk record -> ids
// [-> nnorm] -> rows) with the row loads of one round waited for before the next round is issued: a walker keeps ~8 gathers of
// 160 bytes in flight a third of the time, where the chip needs ~8 per walker ALL the time (8 TB/s x 2.5 us / 256 CUs / 64
// walkers).  What it needs: the walker's chunks as one contiguous edge stream with the ids two windows and the rows one round
// ahead of the products -- tried (contiguous chunk range per walker, prefetched id / record windows, two alternating sets of
// eight gathers): bit-identical, but 4.6 ms: with loads inside the data-dependent chunk-retire path the compiler cannot count
// outstanding loads and drains them all (s_waitcnt vmcnt(0)) at the top of every window and at every record-window switch, so
// the prefetch never overlaps; it needs hand-placed wait counts (inline assembly) or the records delivered through LDS.
#pragma once
#include "rgcn_tile.hpp"

namespace temp {

#define SLICE_THREADS 1024
#define SLICE_LDS_MAX (156 * 1024)

struct SliceArgs { int n_slices, fs4, n_parts, lds_bytes; };

// the fewest slices (widest rows, <= 16 float4) whose table slice fits; the 32 block slots of an XCD = n_slices x n_parts
inline bool slice_plan(int D, int S, int n_rel_rows, SliceArgs* a) {
  const int D4 = D >> 2;
  for (int ns = ceil_div(D4, 16); ns <= D4 && ns <= 32; ++ns) {
    const int fs4 = ceil_div(D4, ns);
    const size_t bytes = (size_t)n_rel_rows * S * fs4 * 16;
    if (bytes > SLICE_LDS_MAX) continue;
    a->n_slices = ns; a->fs4 = fs4; a->n_parts = 32 / ns; a->lds_bytes = (int)bytes;
    return a->n_parts >= 1;
  }
  return false;
}

// lane of walker-local index e (0..15) in the 32-lane half of `lane` that holds walker group `grp` (inverse of tile_lane)
__device__ __forceinline__ int slice_lane_of(int lane, int grp, int e) {
  int l;
  if (grp == 0) l = e < 4 ? e : (e < 8 ? e + 8 : e + 12);
  else l = e < 8 ? e + 4 : (e < 12 ? e + 8 : e + 16);
  return (lane & 32) | l;
}

template <int S, int MODE>
__global__ void __launch_bounds__(SLICE_THREADS) k_rgcn_agg_f(TempEdgeView v, SliceArgs sa, const float* __restrict__ feat, int ldf,
                                                              const int32_t* __restrict__ feat_ids, const float* __restrict__ W, int n_rel_rows,
                                                              const float* __restrict__ nnorm, int D, float* __restrict__ out,
                                                              float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float4 slice_ws[];
  const int D4 = D >> 2;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int slice = local % sa.n_slices, part = local / sa.n_slices;
  if (part >= sa.n_parts) return;                              // (uniform) slots beyond n_slices x n_parts stay empty
  const int f4_0 = (slice * D4) / sa.n_slices, nf4 = ((slice + 1) * D4) / sa.n_slices - f4_0, fs4 = sa.fs4;
  const int tid = threadIdx.x;
  {  // table slice: Ws[(r * S + j) * fs4 + lr] = W4[r * D4 * S + (f4_0 + lr) * S + j]
    const float4* W4 = reinterpret_cast<const float4*>(W);
    const int total = n_rel_rows * S * fs4;
    for (int q = tid; q < total; q += SLICE_THREADS) {
      const int rj = q / fs4, lr = q - rj * fs4;
      const int r = rj / S, j = rj - r * S;
      if (lr < nf4) slice_ws[q] = W4[(size_t)r * D4 * S + (size_t)(f4_0 + lr) * S + j];
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  int g, lr;
  tile_lane(lane, g, lr);
  const int grp = g & 1;
  const bool lane_ok = lr < nf4;
  const unsigned char* wl = reinterpret_cast<const unsigned char*>(slice_ws + lr);
  const unsigned xrow = (unsigned)fs4 * 16u, wrow_b = (unsigned)(S * fs4) * 16u;
  const float* fcol = feat + (size_t)(f4_0 + (lane_ok ? lr : 0)) * 4;
  int c_lo, c_hi;                                             // XCD x: the chunks of the x-th eighth of the EDGES (common.hpp)
  xcd_chunk_range(v.n_chunks, v.n_edges, v.chunk_beg, xcd, c_lo, c_hi);
  constexpr int WAVES = SLICE_THREADS / 64;
  for (int c0 = c_lo + 4 * (part * WAVES + wave); c0 < c_hi; c0 += 4 * WAVES * sa.n_parts) {
    const int c = c0 + g;
    const bool has = c < c_hi;
    int seg = 0, beg = 0, cnt = 0, slot = -1;
    if (has) { seg = v.chunk_seg[c]; beg = v.chunk_beg[c]; cnt = v.chunk_end[c] - beg; slot = v.chunk_slot[c]; }
    float4 acc = zero4();
    for (int j0 = 0; j0 < cnt; j0 += 16) {                     // (per-walker trip count)
      int a_l = 0, b_l = 0;
      float s_l = 1.f;
      if (j0 + lr < cnt) {                                     // lane lr of the walker: edge j0 + lr
        a_l = v.a[beg + j0 + lr];
        b_l = v.b[beg + j0 + lr];
        if (MODE == MODE_DX) { const float nn = nnorm[a_l]; s_l = nn * nn; }
        if (feat_ids) a_l = feat_ids[a_l];
      }
      const int n16 = min(16, cnt - j0);
      constexpr int U = 8;
      for (int e0 = 0; e0 < n16; e0 += U) {
        float4 x[U];
        int rel[U];
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int src_lane = slice_lane_of(lane, grp, (e0 + u) & 15);
          const int row = __shfl(a_l, src_lane);
          rel[u] = __shfl(b_l, src_lane);
          sc[u] = __shfl(s_l, src_lane);
          x[u] = ld4(fcol + (size_t)row * ldf);             // (unconditional: a select on the loaded value makes the wave wait for every
                                                              //  gather before it issues the next; idle lanes / edges read a valid row and are not used)
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (e0 + u < n16) {
            const unsigned char* wr = wl + __umul24((unsigned)rel[u], wrow_b);
            float4 w[S];
#pragma unroll
            for (int q = 0; q < S; ++q) w[q] = *reinterpret_cast<const float4*>(wr + q * xrow);
            block_mac<S, MODE>(acc, x[u], w, sc[u]);
          }
      }
    }
    if (has && lane_ok) {
      if (MODE == MODE_FWD) { const float nn = nnorm[seg]; acc = scale4(acc, nn * nn); }
      float* dst = (slot < 0) ? out + (size_t)seg * D : partial + (size_t)slot * D;
      st4(dst + (f4_0 + lr) * 4, acc);
    }
  }
}

}  // namespace temp
