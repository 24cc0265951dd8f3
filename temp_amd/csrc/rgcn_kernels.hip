// RGCN message passing for gfx950: chunked segmented reduce over sorted edge views.
//
// Replaces RGCNLayer.msg_func / propagate / apply_func (models/RGCN.py:91-104 of the TeMP
// reference) and their autograd.  One wave owns one chunk (<= 64 edges of a single segment).
// Lanes are split into 64/LPR groups of LPR lanes; a group owns one edge at a time and each lane
// of the group owns 4 consecutive features (float4), so a feature row is one coalesced
// LPR*16-byte read.  The per-relation block-diagonal weight row is read from LDS when the whole
// table fits in 64 KB (GDELT: 40 rows x 1600 B), else through L2.  No atomics: segments that span
// several chunks go through ordered partial slots + a fix-up pass, so results are deterministic.
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "common.hpp"
#include "side_stream.hpp"

namespace temp {

enum { MODE_FWD = 0, MODE_DX = 1 };

// acc += x . BD-block(s) (MODE_FWD) or acc += x . BD-block(s)^T (MODE_DX) for the 4 features of a lane.
template <int S, int MODE>
__device__ __forceinline__ void block_mac(float4& acc, const float4 x, const float4* w, const float c) {
  if (S == 1) {
    acc.x = fmaf(c * x.x, w[0].x, acc.x);
    acc.y = fmaf(c * x.y, w[0].y, acc.y);
    acc.z = fmaf(c * x.z, w[0].z, acc.z);
    acc.w = fmaf(c * x.w, w[0].w, acc.w);
  } else if (S == 2) {
    // w[0] = block b: (w00 w01 w10 w11), w[1] = block b+1
    const float x0 = c * x.x, x1 = c * x.y, x2 = c * x.z, x3 = c * x.w;
    if (MODE == MODE_FWD) {
      acc.x = fmaf(x0, w[0].x, fmaf(x1, w[0].z, acc.x));
      acc.y = fmaf(x0, w[0].y, fmaf(x1, w[0].w, acc.y));
      acc.z = fmaf(x2, w[1].x, fmaf(x3, w[1].z, acc.z));
      acc.w = fmaf(x2, w[1].y, fmaf(x3, w[1].w, acc.w));
    } else {
      acc.x = fmaf(x0, w[0].x, fmaf(x1, w[0].y, acc.x));
      acc.y = fmaf(x0, w[0].z, fmaf(x1, w[0].w, acc.y));
      acc.z = fmaf(x2, w[1].x, fmaf(x3, w[1].y, acc.z));
      acc.w = fmaf(x2, w[1].z, fmaf(x3, w[1].w, acc.w));
    }
  } else {  // S == 4: w[i] = row i of the 4x4 block
    const float4 xs = scale4(x, c);
    if (MODE == MODE_FWD) {
      acc = fma4(xs.x, w[0], acc);
      acc = fma4(xs.y, w[1], acc);
      acc = fma4(xs.z, w[2], acc);
      acc = fma4(xs.w, w[3], acc);
    } else {
      acc.x += dot4(xs, w[0]);
      acc.y += dot4(xs, w[1]);
      acc.z += dot4(xs, w[2]);
      acc.w += dot4(xs, w[3]);
    }
  }
}

// Stage the weight table into LDS, permuted so that the j-th float4 of every lane of a feature row
// is contiguous (conflict-free ds_read_b128):  Ws4[(r*S + j)*D4 + lr]  <-  W4[r*D4*S + lr*S + j].
template <int S>
__device__ __forceinline__ void stage_weights(float4* Ws4, const float* W, int n_rows, int D4) {
  const int total = n_rows * D4 * S;
  const float4* W4 = reinterpret_cast<const float4*>(W);
  for (int q = threadIdx.x; q < total; q += blockDim.x) {
    const int r = q / (D4 * S);
    const int rem = q - r * (D4 * S);
    const int lr = rem / S, j = rem - lr * S;
    Ws4[(r * S + j) * D4 + lr] = W4[q];
  }
}

// One kernel for the forward aggregation (view = by-dst, a = src, b = rel, post-scale nnorm[seg]^2)
// and for d/dh (view = by-src, a = dst, b = rel, per-edge scale nnorm[dst]^2, transposed blocks).
template <int S, int MODE, bool W_LDS>
__global__ void __launch_bounds__(1024) k_rgcn_agg(TempEdgeView v, const float* __restrict__ feat, int ldf,
                                                   const int32_t* __restrict__ feat_ids, const float* __restrict__ W,
                                                   int n_rel_rows, const float* __restrict__ nnorm, int D, int lpr,
                                                   float* __restrict__ out, float* __restrict__ partial) {
  extern __shared__ float4 Ws4[];
  const int D4 = D >> 2;
  if (W_LDS) {
    stage_weights<S>(Ws4, W, n_rel_rows, D4);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int lr = lane & (lpr - 1), gi = lane / lpr, epw = 64 / lpr;
  const int f = lr << 2;
  const bool active = f < D;
  ItemRange it = xcd_chunks(v.n_chunks, v.n_edges, v.chunk_beg, wpb);
  for (int c = it.beg + wave; c < it.end; c += it.stride) {
    const int seg = v.chunk_seg[c], beg = v.chunk_beg[c], cnt = v.chunk_end[c] - beg, slot = v.chunk_slot[c];
    int a_l = 0, b_l = 0;
    float s_l = 1.f;
    if (lane < cnt) {
      a_l = v.a[beg + lane];
      b_l = v.b[beg + lane];
      if (MODE == MODE_DX) { const float nn = nnorm[a_l]; s_l = nn * nn; }
      if (feat_ids) a_l = feat_ids[a_l];
    }
    float4 acc = zero4();
    constexpr int U = 4;
    for (int e0 = 0; e0 < cnt; e0 += epw * U) {
      float4 xv[U];
      int rr[U];
      float sc[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * epw + gi;
        ok[u] = (e < cnt) && active;
        const int row = __shfl(a_l, e & 63);
        rr[u] = __shfl(b_l, e & 63);
        sc[u] = __shfl(s_l, e & 63);
        xv[u] = ok[u] ? ld4(feat + (size_t)row * ldf + f) : zero4();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) {
          float4 w[S];
          if (W_LDS) {
#pragma unroll
            for (int j = 0; j < S; ++j) w[j] = Ws4[(rr[u] * S + j) * D4 + lr];
          } else {
            const float* wr = W + (size_t)rr[u] * (D * S) + f * S;
#pragma unroll
            for (int j = 0; j < S; ++j) w[j] = ld4(wr + 4 * j);
          }
          block_mac<S, MODE>(acc, xv[u], w, sc[u]);
        }
      }
    }
    for (int off = lpr; off < 64; off <<= 1) acc = add4(acc, shfl_xor4(acc, off));
    if (gi == 0 && active) {
      if (MODE == MODE_FWD) { const float nn = nnorm[seg]; acc = scale4(acc, nn * nn); }
      float* dst = (slot < 0) ? out + (size_t)seg * D + f : partial + (size_t)slot * D + f;
      st4(dst, acc);
    }
  }
}

// ---- wide rows (128 < D <= 256: one edge per wave pass) --------------------------------------------------------------------
// The kernels above spend ~45 instructions per edge (PMC / ISA count: the S-gdelt launches are bound by instruction ISSUE, not by
// the L2 gathers): lane permutes for the edge's ids, zero-fill moves, exec-mask bookkeeping of the `ok` branches, 64-bit vector
// address arithmetic.  With one edge per pass every per-edge quantity is wave-uniform, so it lives in SCALAR registers here:
// v_readlane of the pre-loaded ids, scalar row base + one lane offset for the gather (no vector address math), scalar loop
// control, no per-edge branches (full groups of four, then a scalar tail).  ~20 vector instructions per edge.
template <int S, int MODE, bool W_LDS>
__global__ void __launch_bounds__(1024, 8) k_rgcn_agg_s(TempEdgeView v, const float* __restrict__ feat, int ldf,
                                                     const int32_t* __restrict__ feat_ids, const float* __restrict__ W, int n_rel_rows,
                                                     const float* __restrict__ nnorm, int D, float* __restrict__ out,
                                                     float* __restrict__ partial) {
  extern __shared__ float4 Ws4[];
  const int D4 = D >> 2;
  if (W_LDS) {
    stage_weights<S>(Ws4, W, n_rel_rows, D4);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int f = lane << 2;
  const bool active = lane < D4;
  ItemRange it = xcd_chunks(v.n_chunks, v.n_edges, v.chunk_beg, wpb);
  for (int c = it.beg + wave; c < it.end; c += it.stride) {
    const int seg = __builtin_amdgcn_readfirstlane(v.chunk_seg[c]), beg = __builtin_amdgcn_readfirstlane(v.chunk_beg[c]);
    const int cnt = __builtin_amdgcn_readfirstlane(v.chunk_end[c]) - beg, slot = __builtin_amdgcn_readfirstlane(v.chunk_slot[c]);
    int a_l = 0, b_l = 0;
    float s_l = 1.f;
    if (lane < cnt) {
      a_l = v.a[beg + lane];
      b_l = v.b[beg + lane];
      if (MODE == MODE_DX) { const float nn = nnorm[a_l]; s_l = nn * nn; }
      if (feat_ids) a_l = feat_ids[a_l];
    }
    // Relation runs (table beyond LDS only).  The views list a segment's edges in relation order (host planner), so a HUB's chunks
    // are a few long runs of one relation each -- with power-law degrees half of all edges sit in such chunks.  A run needs its
    // 1.6 KB of block weights ONCE (they were read through L2 per edge: twice the bytes of the row itself, and what bounded the
    // 230-relation case), and by linearity its rows are summed first and multiplied once.
    unsigned long long starts = 0ull;
    // (only for tables beyond LDS, n_rel_rows * D * S * 4 > 64 KB: small tables keep the per-edge sums of the LDS-resident kernels)
    if (!W_LDS && (size_t)n_rel_rows * D * S * sizeof(float) > 65536) {
      const int prev = __shfl_up(b_l, 1);
      starts = __builtin_amdgcn_ballot_w64(lane < cnt && (lane == 0 || b_l != prev));
    }
    if (!active) continue;                                     // (lanes past the row keep out of the loads; readlane ignores exec)
    float4 acc = zero4();
    const float4* wl = Ws4 + lane;
    if (!W_LDS && starts != 0ull && 2 * __builtin_popcountll(starts) <= cnt) {  // runs of two and more on average: the run walk
      int e = 0;
      while (e < cnt) {
        const unsigned long long later = e < 63 ? (starts >> (e + 1)) << (e + 1) : 0ull;
        const int end = later ? __builtin_ctzll(later) : cnt;
        const int rel = __builtin_amdgcn_readlane(b_l, e);
        const float* wr = W + (size_t)rel * (D * S) + f * S;
        float4 w[S];
#pragma unroll
        for (int j = 0; j < S; ++j) w[j] = ld4(wr + 4 * j);
        float4 xs = zero4();
        auto row = [&](int i) {
          const int r = __builtin_amdgcn_readlane(a_l, i);
          return ld4(feat + (size_t)r * ldf + f);
        };
        auto scale = [&](int i) { return MODE == MODE_DX ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s_l), i)) : 1.f; };
        int i = e;
        for (; i + 4 <= end; i += 4) {
          const float4 x0 = row(i), x1 = row(i + 1), x2 = row(i + 2), x3 = row(i + 3);
          if (MODE == MODE_DX) { xs = fma4(scale(i), x0, xs); xs = fma4(scale(i + 1), x1, xs); xs = fma4(scale(i + 2), x2, xs); xs = fma4(scale(i + 3), x3, xs); }
          else xs = add4(add4(xs, x0), add4(add4(x1, x2), x3));
        }
        for (; i < end; ++i) {
          const float4 x0 = row(i);
          if (MODE == MODE_DX) xs = fma4(scale(i), x0, xs); else xs = add4(xs, x0);
        }
        block_mac<S, MODE>(acc, xs, w, 1.f);
        e = end;
      }
      if (MODE == MODE_FWD) { const float nn = nnorm[seg]; acc = scale4(acc, nn * nn); }
      float* dst = (slot < 0) ? out + (size_t)seg * D + f : partial + (size_t)slot * D + f;
      st4(dst, acc);
      continue;
    }
    auto edge = [&](int e, float4& x, int& rel, float& sc) {
      const int row = __builtin_amdgcn_readlane(a_l, e);
      rel = __builtin_amdgcn_readlane(b_l, e);
      sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s_l), e));
      x = ld4(feat + (size_t)row * ldf + f);
    };
    auto mac = [&](const float4 x, int rel, float sc) {
      float4 w[S];
      if (W_LDS) {
#pragma unroll
        for (int j = 0; j < S; ++j) w[j] = wl[(rel * S + j) * D4];
      } else {                                                 // the relation table does not fit LDS: scalar row base, through L2
        const float* wr = W + (size_t)rel * (D * S) + f * S;
#pragma unroll
        for (int j = 0; j < S; ++j) w[j] = ld4(wr + 4 * j);
      }
      block_mac<S, MODE>(acc, x, w, sc);
    };
    int e = 0;
    for (; e + 4 <= cnt; e += 4) {
      float4 x0, x1, x2, x3;
      int r0, r1, r2, r3;
      float c0, c1, c2, c3;
      edge(e, x0, r0, c0); edge(e + 1, x1, r1, c1); edge(e + 2, x2, r2, c2); edge(e + 3, x3, r3, c3);
      mac(x0, r0, c0); mac(x1, r1, c1); mac(x2, r2, c2); mac(x3, r3, c3);
    }
    for (; e < cnt; ++e) {
      float4 x0;
      int r0;
      float c0;
      edge(e, x0, r0, c0);
      mac(x0, r0, c0);
    }
    if (MODE == MODE_FWD) { const float nn = nnorm[seg]; acc = scale4(acc, nn * nn); }
    float* dst = (slot < 0) ? out + (size_t)seg * D + f : partial + (size_t)slot * D + f;
    st4(dst, acc);
  }
}

template <int S>
__global__ void __launch_bounds__(256) k_rgcn_dw_s(TempEdgeView v, const float* __restrict__ x, const int32_t* __restrict__ x_ids,
                                                   const float* __restrict__ dz, const float* __restrict__ nnorm, int D,
                                                   float* __restrict__ dW, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int f = lane << 2;
  const bool active = lane < (D >> 2);
  const int wrow = D * S;
  ItemRange it = xcd_chunks(v.n_chunks, v.n_edges, v.chunk_beg, wpb);
  for (int c = it.beg + wave; c < it.end; c += it.stride) {
    const int seg = __builtin_amdgcn_readfirstlane(v.chunk_seg[c]), cbeg = __builtin_amdgcn_readfirstlane(v.chunk_beg[c]);
    const int cend = __builtin_amdgcn_readfirstlane(v.chunk_end[c]), slot = __builtin_amdgcn_readfirstlane(v.chunk_slot[c]);
    float4 acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = zero4();
    for (int beg = cbeg; beg < cend; beg += 64) {             // a relation chunk holds up to TEMP_CHUNK_REL edges
      const int cnt = min(64, cend - beg);
      int a_l = 0, b_l = 0;
      float s_l = 0.f;
      if (lane < cnt) {
        a_l = v.a[beg + lane];
        if (x_ids) a_l = x_ids[a_l];                          // x is a table, the node's row is x[x_ids[node]]
        b_l = v.b[beg + lane];
        const float nn = nnorm[b_l];
        s_l = nn * nn;
      }
      // Destination runs: the view lists a relation's edges in destination order, so the edges into a hub are one run -- its dz
      // row (and norm) is read ONCE and, by linearity, the run's source rows are summed before the outer product.
      unsigned long long starts = 0ull;
      {
        const int prev = __shfl_up(b_l, 1);
        starts = __builtin_amdgcn_ballot_w64(lane < cnt && (lane == 0 || b_l != prev));
      }
      if (!active) continue;
      auto mac = [&](const float4 xx, const float4 g) {
        if (S == 1) {
          acc[0].x = fmaf(xx.x, g.x, acc[0].x);
          acc[0].y = fmaf(xx.y, g.y, acc[0].y);
          acc[0].z = fmaf(xx.z, g.z, acc[0].z);
          acc[0].w = fmaf(xx.w, g.w, acc[0].w);
        } else if (S == 2) {
          acc[0].x = fmaf(xx.x, g.x, acc[0].x);  // blk0 w00
          acc[0].y = fmaf(xx.x, g.y, acc[0].y);  //      w01
          acc[0].z = fmaf(xx.y, g.x, acc[0].z);  //      w10
          acc[0].w = fmaf(xx.y, g.y, acc[0].w);  //      w11
          acc[1].x = fmaf(xx.z, g.z, acc[1].x);  // blk1
          acc[1].y = fmaf(xx.z, g.w, acc[1].y);
          acc[1].z = fmaf(xx.w, g.z, acc[1].z);
          acc[1].w = fmaf(xx.w, g.w, acc[1].w);
        } else {
          acc[0] = fma4(xx.x, g, acc[0]);
          acc[1] = fma4(xx.y, g, acc[1]);
          acc[2] = fma4(xx.z, g, acc[2]);
          acc[3] = fma4(xx.w, g, acc[3]);
        }
      };
      if (2 * __builtin_popcountll(starts) <= cnt) {          // runs of two and more on average: the run walk
        int e = 0;
        while (e < cnt) {
          const unsigned long long later = e < 63 ? (starts >> (e + 1)) << (e + 1) : 0ull;
          const int end = later ? __builtin_ctzll(later) : cnt;
          const int dst = __builtin_amdgcn_readlane(b_l, e);
          const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s_l), e));
          const float4 g = scale4(ld4(dz + (size_t)dst * D + f), sc);
          auto row = [&](int i) { return ld4(x + (size_t)__builtin_amdgcn_readlane(a_l, i) * D + f); };
          float4 xs = zero4();
          int i = e;
          for (; i + 4 <= end; i += 4) {
            const float4 x0 = row(i), x1 = row(i + 1), x2 = row(i + 2), x3 = row(i + 3);
            xs = add4(add4(xs, x0), add4(add4(x1, x2), x3));
          }
          for (; i < end; ++i) xs = add4(xs, row(i));
          mac(xs, g);
          e = end;
        }
        continue;
      }
      auto edge = [&](int e, float4& xx, float4& g) {
        const int src = __builtin_amdgcn_readlane(a_l, e), dst = __builtin_amdgcn_readlane(b_l, e);
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s_l), e));
        xx = ld4(x + (size_t)src * D + f);
        g = scale4(ld4(dz + (size_t)dst * D + f), sc);
      };
      int e = 0;
      for (; e + 4 <= cnt; e += 4) {
        float4 x0, x1, x2, x3, g0, g1, g2, g3;
        edge(e, x0, g0); edge(e + 1, x1, g1); edge(e + 2, x2, g2); edge(e + 3, x3, g3);
        mac(x0, g0); mac(x1, g1); mac(x2, g2); mac(x3, g3);
      }
      for (; e < cnt; ++e) {
        float4 x0, g0;
        edge(e, x0, g0);
        mac(x0, g0);
      }
    }
    if (active) {
      float* dst = ((slot < 0) ? dW + (size_t)seg * wrow : partial + (size_t)slot * wrow) + f * S;
#pragma unroll
      for (int j = 0; j < S; ++j) st4(dst + 4 * j, acc[j]);
    }
  }
}

// ordered sum of `end - s` partial rows (column `p`), eight loads in flight: THE summation order of every fix-up
__device__ __forceinline__ float4 fixup_walk(const float* __restrict__ p, int s, int end, int width) {
  float4 acc = zero4();
  for (; s + 8 <= end; s += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld4(p + (size_t)(s + u) * width);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
  }
  for (; s < end; ++s) acc = add4(acc, ld4(p + (size_t)s * width));
  return acc;
}

}  // namespace temp
#include "rgcn_tile.hpp"
namespace temp {

// Generic (any si, so) scalar-lane variant; slow, for shapes outside the fast path.
template <int MODE>
__global__ void __launch_bounds__(256) k_rgcn_agg_generic(TempEdgeView v, const float* __restrict__ feat, int ldf,
                                                          const int32_t* __restrict__ feat_ids, const float* __restrict__ W,
                                                          const float* __restrict__ nnorm, int d_in, int d_out, int si, int so,
                                                          float* __restrict__ out, float* __restrict__ partial) {
  // MODE_FWD: result width d_out, input width d_in.  MODE_DX: result width d_in, input (grad) width d_out.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int wres = (MODE == MODE_FWD) ? d_out : d_in;
  const int wrow = (d_in / si) * si * so;
  for (int c = blockIdx.x * wpb + wave; c < v.n_chunks; c += gridDim.x * wpb) {
    const int seg = v.chunk_seg[c], beg = v.chunk_beg[c], end = v.chunk_end[c], slot = v.chunk_slot[c];
    for (int o = lane; o < wres; o += 64) {
      float acc = 0.f;
      for (int e = beg; e < end; ++e) {
        int row = v.a[e];
        const int r = v.b[e];
        float c2 = 1.f;
        if (MODE == MODE_DX) { const float nn = nnorm[row]; c2 = nn * nn; }
        if (feat_ids) row = feat_ids[row];
        const float* x = feat + (size_t)row * ldf;
        const float* w = W + (size_t)r * wrow;
        if (MODE == MODE_FWD) {
          const int b = o / so, oo = o - b * so;
          for (int i = 0; i < si; ++i) acc = fmaf(x[b * si + i], w[b * si * so + i * so + oo], acc);
        } else {
          const int b = o / si, ii = o - b * si;
          float t = 0.f;
          for (int q = 0; q < so; ++q) t = fmaf(x[b * so + q], w[b * si * so + ii * so + q], t);
          acc = fmaf(c2, t, acc);
        }
      }
      if (MODE == MODE_FWD) { const float nn = nnorm[seg]; acc *= nn * nn; }
      float* dst = (slot < 0) ? out + (size_t)seg * wres : partial + (size_t)slot * wres;
      dst[o] = acc;
    }
  }
}

// out[seg] = sum of its partial slots, in a fixed order (deterministic).  One wave per (fix entry, 256-float column block);
// lanes own float4s; the slot walk is unrolled for memory-level parallelism.  A segment with more than LONG partial rows
// (a hub of an HBM-sized snapshot has tens of thousands: one wave would walk them for a millisecond while the chip idles) is
// walked by ALL waves of its block, each over a contiguous share of the slots, and the shares are added in wave order.
// <WAVES, IPB, LONG>: waves per block, items a block takes per round (one per wave, IPB <= WAVES), partial rows from which an
// entry counts as long.  Many entries (the node views: thousands of hubs with a few partial rows each): <4, 4, 256>.  Few entries
// (the by-relation view of the weight gradient: one entry per relation, hundreds to thousands of partial rows each -- ten
// 4-wave blocks walked them for 40 us on an idle chip): <16, 1, 32>, one block of 16 waves per entry.
constexpr int FIX_PART = 256, FIX_LIST = 2, FIX_LIST_CAP = 4096, FIX_SPLIT_MIN = 1 << 15;
// SPLIT: an entry with more than `huge` partial rows is not summed here; its index is appended to the list in `ctl` and
// k_fixup_split sums it with many blocks (the append order varies from run to run, the sums do not: see there).
template <int WAVES, int IPB, int LONG, bool SPLIT = false>
__global__ void __launch_bounds__(WAVES * 64) k_fixup(int n_fix, const int32_t* __restrict__ fix_seg, const int32_t* __restrict__ fix_slot,
                                                      const int32_t* __restrict__ fix_cnt, const float* __restrict__ partial, int width,
                                                      float* __restrict__ out, unsigned* __restrict__ ctl = nullptr, int huge = 0) {
  __shared__ long long long_item[IPB];
  __shared__ float4 share[WAVES][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cblocks = (width + 255) >> 8;
  const long long items = (long long)n_fix * cblocks;
  const long long rounds = (items + (long long)gridDim.x * IPB - 1) / ((long long)gridDim.x * IPB);
  for (long long r = 0; r < rounds; ++r) {                       // block-uniform
    const long long it = (r * gridDim.x + blockIdx.x) * IPB + wave;
    long long mine = -1;
    if (wave < IPB && it < items) {
      const int i = (int)(it / cblocks), cb = (int)(it - (long long)i * cblocks);
      const int f = (cb << 8) + (lane << 2);
      const int seg = fix_seg[i], s0 = fix_slot[i], cnt = fix_cnt[i];          // (one round trip for the three)
      if (SPLIT && cnt > huge) {                                // left to k_fixup_split (once per entry: by its first column block)
        if (cb == 0 && lane == 0) ctl[FIX_LIST + atomicAdd(ctl, 1u)] = (unsigned)i;
      } else if (cnt > LONG) mine = it;                         // left to the whole block below
      else if (f < width) st4(out + (size_t)seg * width + f, fixup_walk(partial + (size_t)s0 * width + f, 0, cnt, width));
    }
    if (wave < IPB && lane == 0) long_item[wave] = mine;
    if (!__syncthreads_or(mine >= 0)) continue;                 // (block-uniform) no long entry in this round: nothing is shared
    for (int w = 0; w < IPB; ++w) {
      const long long lt = long_item[w];                        // block-uniform
      if (lt < 0) continue;
      const int i = (int)(lt / cblocks), cb = (int)(lt - (long long)i * cblocks);
      const int f = (cb << 8) + (lane << 2);
      const int seg = fix_seg[i], slot0 = fix_slot[i], cnt = fix_cnt[i];
      const int per = (cnt + WAVES - 1) / WAVES;
      const int s0 = min(cnt, wave * per), s1 = min(cnt, s0 + per);
      share[wave][lane] = f < width ? fixup_walk(partial + (size_t)slot0 * width + f, s0, s1, width) : zero4();
      __syncthreads();
      if (wave == 0 && f < width) {
        float4 acc = share[0][lane];
#pragma unroll
        for (int u = 1; u < WAVES; ++u) acc = add4(acc, share[u][lane]);
        st4(out + (size_t)seg * width + f, acc);
      }
      __syncthreads();
    }
    __syncthreads();                                            // long_item is rewritten by the next round
  }
}

// Second level of the fix-up for entries with thousands of partial rows (a hub of an HBM-sized snapshot: 18 000; a relation of
// a 20-relation graph of that size: 3 300 rows of the weight gradient) -- one block, however wide, walks such an entry for
// hundreds of microseconds while the chip idles.  The listed entries are cut into parts of FIX_PART rows; (entry, part) pairs are
// dealt to all blocks; a part's sum goes to its scratch row; the block that finishes an entry's LAST outstanding part (a ticket
// per entry) adds the entry's part rows IN PART ORDER.  Which block that is, and the order of the list, vary from run to run; the
// sum of a part and the order of the final addition do not, so the result is deterministic.
// ctl words: [0] listed entries, [1] unused, [FIX_LIST + j] entry index, [FIX_LIST + cap + j * cblocks + cb] ticket of list position j,
// column block cb.
__host__ __device__ inline int fix_huge(int n_partial) { const int h = (n_partial + FIX_LIST_CAP - 2) / (FIX_LIST_CAP - 1); return h > 2048 ? h : 2048; }
inline int fix_list_cap(int n_partial) { return n_partial / fix_huge(n_partial) + 1; }          // (every listed entry has > huge rows)
inline size_t fix_ctl_bytes(int n_partial, size_t width) { return align_up((size_t)(FIX_LIST + (1 + (width + 255) / 256) * fix_list_cap(n_partial)) * sizeof(unsigned), 256); }
inline size_t fix_scratch_rows(int n_partial) { return (size_t)n_partial / FIX_PART + fix_list_cap(n_partial); }
// bytes of a view's partial region: the slots, then (large views only) control words + part rows of the split fix-up
inline size_t partial_bytes(int n_partial, size_t width) {
  size_t b = align_up((size_t)n_partial * width * sizeof(float), 256);
  if (n_partial >= FIX_SPLIT_MIN) b += fix_ctl_bytes(n_partial, width) + align_up(fix_scratch_rows(n_partial) * width * sizeof(float), 256);
  return b;
}

__global__ void __launch_bounds__(256) k_fixup_split(const int32_t* __restrict__ fix_seg, const int32_t* __restrict__ fix_slot,
                                                     const int32_t* __restrict__ fix_cnt, const float* __restrict__ partial, int width,
                                                     float* __restrict__ out, unsigned* __restrict__ ctl, int cap, float* __restrict__ scratch) {
  __shared__ int pre[FIX_LIST_CAP + 1];
  __shared__ float4 share[4][64];
  __shared__ int last_flag;
  const int n_h = (int)ctl[0];
  if (n_h == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cblocks = (width + 255) >> 8;
  // parts of every listed entry, then their exclusive prefix (one thread: at most a few thousand short additions)
  for (int j = tid; j < n_h; j += blockDim.x) pre[j + 1] = (fix_cnt[ctl[FIX_LIST + j]] + FIX_PART - 1) / FIX_PART;
  __syncthreads();
  if (tid == 0) { pre[0] = 0; int run = 0; for (int j = 1; j <= n_h; ++j) { run += pre[j]; pre[j] = run; } }
  __syncthreads();
  const int total = pre[n_h] * cblocks;
  unsigned* ticket = ctl + FIX_LIST + cap;
  for (int w = blockIdx.x; w < total; w += gridDim.x) {                      // block-uniform
    const int pw = w / cblocks, cb = w - pw * cblocks;
    int lo = 0, hi = n_h;                                                    // list position j with pre[j] <= pw < pre[j + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= pw) lo = mid; else hi = mid; }
    const int j = lo, part = pw - pre[j], parts = pre[j + 1] - pre[j];
    const int i = (int)ctl[FIX_LIST + j];
    const int seg = fix_seg[i], slot0 = fix_slot[i], cnt = fix_cnt[i];
    const int f = (cb << 8) + (lane << 2);
    const int p0 = part * FIX_PART, p1 = min(cnt, p0 + FIX_PART);
    const int per = (p1 - p0 + 3) >> 2;
    const int s0 = min(p1, p0 + wave * per), s1 = min(p1, s0 + per);
    share[wave][lane] = f < width ? fixup_walk(partial + (size_t)slot0 * width + f, s0, s1, width) : zero4();
    __syncthreads();
    if (wave == 0 && f < width)
      st4(scratch + (size_t)(pre[j] + part) * width + f, add4(add4(share[0][lane], share[1][lane]), add4(share[2][lane], share[3][lane])));
    __threadfence();                                                         // the part row is visible chip-wide before the ticket moves
    __syncthreads();
    if (tid == 0) last_flag = atomicAdd(&ticket[j * cblocks + cb], 1u) == (unsigned)(parts - 1);
    __syncthreads();
    if (last_flag) {                                                         // block-uniform: every part row of (entry, cb) is written
      __threadfence();
      if (wave == 0 && f < width) st4(out + (size_t)seg * width + f, fixup_walk(scratch + (size_t)pre[j] * width + f, 0, parts, width));
    }
    __syncthreads();
  }
}

// d/dweight: view = by-rel (a = src node, b = dst node).  Per edge the lane takes its 4 features of
// x[src] and of c*dz[dst] and accumulates the S x S outer products of its blocks in registers.
template <int S>
__global__ void __launch_bounds__(256) k_rgcn_dw(TempEdgeView v, const float* __restrict__ x, const int32_t* __restrict__ x_ids,
                                                 const float* __restrict__ dz, const float* __restrict__ nnorm, int D, int lpr,
                                                 float* __restrict__ dW, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int lr = lane & (lpr - 1), gi = lane / lpr, epw = 64 / lpr;
  const int f = lr << 2;
  const bool active = f < D;
  const int wrow = D * S;
  ItemRange it = xcd_chunks(v.n_chunks, v.n_edges, v.chunk_beg, wpb);
  for (int c = it.beg + wave; c < it.end; c += it.stride) {
    const int seg = v.chunk_seg[c], cbeg = v.chunk_beg[c], cend = v.chunk_end[c], slot = v.chunk_slot[c];
    float4 acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = zero4();
    // a relation chunk holds up to TEMP_CHUNK_REL edges; walk it 64 edges (one per lane) at a time
    for (int beg = cbeg; beg < cend; beg += 64) {
      const int cnt = min(64, cend - beg);
      int a_l = 0, b_l = 0;
      float s_l = 0.f;
      if (lane < cnt) {
        a_l = v.a[beg + lane];
        if (x_ids) a_l = x_ids[a_l];                      // x is a table, the node's row is x[x_ids[node]]
        b_l = v.b[beg + lane];
        const float nn = nnorm[b_l];
        s_l = nn * nn;
      }
      constexpr int U = 4;
      for (int e0 = 0; e0 < cnt; e0 += epw * U) {
        float4 xv[U], gv[U];
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = e0 + u * epw + gi;
          const bool ok = (e < cnt) && active;
          const int src = __shfl(a_l, e & 63), dst = __shfl(b_l, e & 63);
          sc[u] = __shfl(s_l, e & 63);
          xv[u] = ok ? ld4(x + (size_t)src * D + f) : zero4();
          gv[u] = ok ? ld4(dz + (size_t)dst * D + f) : zero4();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float4 g = scale4(gv[u], sc[u]);
          const float4 xx = xv[u];
          if (S == 1) {
            acc[0].x = fmaf(xx.x, g.x, acc[0].x);
            acc[0].y = fmaf(xx.y, g.y, acc[0].y);
            acc[0].z = fmaf(xx.z, g.z, acc[0].z);
            acc[0].w = fmaf(xx.w, g.w, acc[0].w);
          } else if (S == 2) {
            acc[0].x = fmaf(xx.x, g.x, acc[0].x);  // blk0 w00
            acc[0].y = fmaf(xx.x, g.y, acc[0].y);  //      w01
            acc[0].z = fmaf(xx.y, g.x, acc[0].z);  //      w10
            acc[0].w = fmaf(xx.y, g.y, acc[0].w);  //      w11
            acc[1].x = fmaf(xx.z, g.z, acc[1].x);  // blk1
            acc[1].y = fmaf(xx.z, g.w, acc[1].y);
            acc[1].z = fmaf(xx.w, g.z, acc[1].z);
            acc[1].w = fmaf(xx.w, g.w, acc[1].w);
          } else {
            acc[0] = fma4(xx.x, g, acc[0]);
            acc[1] = fma4(xx.y, g, acc[1]);
            acc[2] = fma4(xx.z, g, acc[2]);
            acc[3] = fma4(xx.w, g, acc[3]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < S; ++j)
      for (int off = lpr; off < 64; off <<= 1) acc[j] = add4(acc[j], shfl_xor4(acc[j], off));
    if (gi == 0 && active) {
      float* dst = ((slot < 0) ? dW + (size_t)seg * wrow : partial + (size_t)slot * wrow) + f * S;
#pragma unroll
      for (int j = 0; j < S; ++j) st4(dst + 4 * j, acc[j]);
    }
  }
}

__global__ void __launch_bounds__(256) k_rgcn_dw_generic(TempEdgeView v, const float* __restrict__ x, const int32_t* __restrict__ x_ids,
                                                         const float* __restrict__ dz,
                                                         const float* __restrict__ nnorm, int d_in, int d_out, int si, int so,
                                                         float* __restrict__ dW, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int wrow = (d_in / si) * si * so;
  for (int c = blockIdx.x * wpb + wave; c < v.n_chunks; c += gridDim.x * wpb) {
    const int seg = v.chunk_seg[c], beg = v.chunk_beg[c], end = v.chunk_end[c], slot = v.chunk_slot[c];
    for (int q = lane; q < wrow; q += 64) {
      const int b = q / (si * so), rem = q - b * si * so, i = rem / so, o = rem - i * so;
      float acc = 0.f;
      for (int e = beg; e < end; ++e) {
        const int src = x_ids ? x_ids[v.a[e]] : v.a[e], dst = v.b[e];
        const float nn = nnorm[dst];
        acc = fmaf(x[(size_t)src * d_in + b * si + i], nn * nn * dz[(size_t)dst * d_out + b * so + o], acc);
      }
      float* d = (slot < 0) ? dW + (size_t)seg * wrow : partial + (size_t)slot * wrow;
      d[q] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// temp_set_option(TEMP_OPT_RGCN_SCALAR, 0) keeps the wide-row launches on the permute-based kernels (A/B runs)
static bool rgcn_scalar_off() { return !option(TEMP_OPT_RGCN_SCALAR); }
static int pick_lpr(int D) {
  int q = D / 4, l = 1;
  while (l < q) l <<= 1;
  return l;
}
static bool fast_shape(int d_in, int d_out, int num_bases, int* S) {
  if (d_in != d_out || d_in % 4 != 0 || d_in > 256 || num_bases <= 0 || d_in % num_bases != 0) return false;
  const int s = d_in / num_bases;
  if (s != 1 && s != 2 && s != 4) return false;
  *S = s;
  return true;
}
static const TempMembers* members_of(const TempGraph* g) { return g->members.n_members > 0 ? &g->members : nullptr; }
static bool view_ok(const TempEdgeView& v) {
  if (v.n_chunks < 0 || v.n_edges < 0 || v.n_partial < 0 || v.n_fix < 0) return false;
  if (v.n_chunks > 0 && (!v.a || !v.b || !v.chunk_seg || !v.chunk_beg || !v.chunk_end || !v.chunk_slot)) return false;
  if (v.n_fix > 0 && (!v.fix_seg || !v.fix_slot || !v.fix_cnt)) return false;
  return true;
}

static bool rgcn_tile_on() { return option(TEMP_OPT_RGCN_TILE) != 0; }
static std::atomic<long long*> g_debug_buf{nullptr};         // development only (temp_set_debug_buffer)
static std::atomic<size_t> g_debug_words{0};
static std::atomic<long long> g_tile_launches{0};           // diagnostic (temp_tile_launches): edge-kernel launches that took the LDS-tiled path

// dynamic LDS beyond 64 KB must be granted per kernel function
template <class K>
static bool tile_grant_lds(K kernel, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}

template <int S, int MODE, class BT>
static bool launch_agg_tile(const TempEdgeView& v, const TileArgs& t, const float* feat, int ldf, const int32_t* ids, const float* W, int n_rel_rows,
                            const float* nnorm, int D, float* out, float* partial, hipStream_t st) {
  static const bool granted = tile_grant_lds(k_rgcn_agg_t<S, MODE, BT>, TILE_LDS_MAX);
  if (!granted) { (void)hipGetLastError(); return false; }
  const int grid = 8 * ceil_div(t.n_members, 8) * t.n_slices;
  TileArgs tp = t;
  if (g_debug_words.load() >= (size_t)grid * 8) tp.prof = g_debug_buf.load();
  if constexpr (S == 2 && MODE == MODE_FWD && sizeof(BT) == 1) {          // development ablations (TEMP_OPT_DEBUG), never set by the product
    const int var = option(TEMP_OPT_DEBUG);
#define TILE_VAR(V) if (var == V) { static const bool ok = tile_grant_lds(k_rgcn_agg_t<S, MODE, BT, V>, TILE_LDS_MAX); (void)ok; \
    TEMP_LAUNCH(K_RGCN_AGG_FWD, (k_rgcn_agg_t<S, MODE, BT, V>), dim3(grid), dim3(TILE_THREADS), t.lds_bytes, st, v, tp, feat, ldf, ids, W, n_rel_rows, nnorm, D, out, partial); return true; }
    TILE_VAR(1) TILE_VAR(2) TILE_VAR(3) TILE_VAR(4) TILE_VAR(7) TILE_VAR(8)
#undef TILE_VAR
  }
  TEMP_LAUNCH((MODE == MODE_FWD ? K_RGCN_AGG_FWD : K_RGCN_AGG_DX), (k_rgcn_agg_t<S, MODE, BT>), dim3(grid), dim3(TILE_THREADS), t.lds_bytes, st, v, tp, feat, ldf, ids,
              W, n_rel_rows, nnorm, D, out, partial);
  g_tile_launches.fetch_add(1, std::memory_order_relaxed);
  return true;
}

template <int S, int MODE>
static bool launch_agg(const TempEdgeView& v, const TempMembers* mb, int view, const float* feat, int ldf, const int32_t* ids, const float* W, int n_rel_rows,
                       const float* nnorm, int D, float* out, float* partial, hipStream_t st) {
  const int lpr = pick_lpr(D);
  const size_t wbytes = (size_t)n_rel_rows * D * S * sizeof(float);
  TileArgs ta;
  if (mb && rgcn_tile_on() && n_rel_rows <= 65535 && v.n_partial < 0xffffff &&
      tile_plan(*mb, view, D, S, 0, n_rel_rows, 0, &ta)) {
    const bool ok = n_rel_rows <= 256 ? launch_agg_tile<S, MODE, unsigned char>(v, ta, feat, ldf, ids, W, n_rel_rows, nnorm, D, out, partial, st)
                                      : launch_agg_tile<S, MODE, unsigned short>(v, ta, feat, ldf, ids, W, n_rel_rows, nnorm, D, out, partial, st);
    if (ok) return ta.fix_off != nullptr;                     // (the tiled kernels then summed the multi-chunk segments themselves)
  }
  if (wbytes <= 65536 && v.n_chunks >= 4096) {
    // whole relation table in LDS; 1024-thread persistent blocks, 2 per CU (2 x 64 KB of 160 KB)
    const int grid = 512;
    if (lpr == 64 && !rgcn_scalar_off())
      TEMP_LAUNCH((MODE == MODE_FWD ? K_RGCN_AGG_FWD : K_RGCN_AGG_DX), (k_rgcn_agg_s<S, MODE, true>), dim3(grid), dim3(1024), wbytes, st, v, feat, ldf, ids, W, n_rel_rows,
                  nnorm, D, out, partial);
    else
      TEMP_LAUNCH((MODE == MODE_FWD ? K_RGCN_AGG_FWD : K_RGCN_AGG_DX), (k_rgcn_agg<S, MODE, true>), dim3(grid), dim3(1024), wbytes, st, v, feat, ldf, ids, W, n_rel_rows, nnorm, D,
                       lpr, out, partial);
  } else {
    int grid = (v.n_chunks + 3) / 4;
    grid = grid < 8 ? 8 : (grid > 2048 ? 2048 : (grid + 7) / 8 * 8);
    // (measured on the S-hbm shape before the relation runs: the forward gains 3 %, d/dh loses 14 %: 3.32 against 2.91 ms; with a
    // table beyond LDS the scalar kernel walks relation runs, which the generic one cannot: TEMP_OPT_DEBUG 101 keeps d/dh generic)
    const bool runs = wbytes > 65536 && option(TEMP_OPT_DEBUG) != 101;
    if (lpr == 64 && (MODE == MODE_FWD || runs) && !rgcn_scalar_off())
      TEMP_LAUNCH((MODE == MODE_FWD ? K_RGCN_AGG_FWD : K_RGCN_AGG_DX), (k_rgcn_agg_s<S, MODE, false>), dim3(grid), dim3(256), 0, st, v, feat, ldf, ids, W, n_rel_rows, nnorm, D,
                  out, partial);
    else
      TEMP_LAUNCH((MODE == MODE_FWD ? K_RGCN_AGG_FWD : K_RGCN_AGG_DX), (k_rgcn_agg<S, MODE, false>), dim3(grid), dim3(256), 0, st, v, feat, ldf, ids, W, n_rel_rows, nnorm, D, lpr,
                       out, partial);
  }
  return false;
}

static void launch_fixup(const TempEdgeView& v, const float* partial, int width, float* out, hipStream_t st) {
  if (v.n_fix <= 0) return;
  long long items = (long long)v.n_fix * ((width + 255) / 256);
  // large views: entries beyond `huge` rows are listed by the first pass and summed by k_fixup_split (partial_bytes() reserved
  // the control words and the part rows behind the slots)
  const bool split = v.n_partial >= FIX_SPLIT_MIN && option(TEMP_OPT_DEBUG) != 100;     // (100: single-level walk, for A/B tests)
  unsigned* ctl = nullptr;
  float* scratch = nullptr;
  int huge = 0, cap = 0;
  if (split) {
    ctl = (unsigned*)((char*)partial + align_up((size_t)v.n_partial * width * sizeof(float), 256));
    scratch = (float*)((char*)ctl + fix_ctl_bytes(v.n_partial, width));
    huge = fix_huge(v.n_partial);
    cap = fix_list_cap(v.n_partial);
    (void)hipMemsetAsync(ctl, 0, fix_ctl_bytes(v.n_partial, width), st);
  }
  if (items <= 1024) {                                          // few entries: one 16-wave block each
    if (split) TEMP_LAUNCH(K_FIXUP, (k_fixup<16, 1, 32, true>), dim3((int)items), dim3(16 * 64), 0, st, v.n_fix, v.fix_seg, v.fix_slot, v.fix_cnt, partial, width, out, ctl, huge);
    else TEMP_LAUNCH(K_FIXUP, (k_fixup<16, 1, 32>), dim3((int)items), dim3(16 * 64), 0, st, v.n_fix, v.fix_seg, v.fix_slot, v.fix_cnt, partial, width, out, ctl, huge);
  } else {
    const long long blocks = (items + 3) / 4;
    int grid = (int)(blocks > 4096 ? 4096 : blocks);
    if (split) TEMP_LAUNCH(K_FIXUP, (k_fixup<4, 4, 256, true>), dim3(grid), dim3(4 * 64), 0, st, v.n_fix, v.fix_seg, v.fix_slot, v.fix_cnt, partial, width, out, ctl, huge);
    else TEMP_LAUNCH(K_FIXUP, (k_fixup<4, 4, 256>), dim3(grid), dim3(4 * 64), 0, st, v.n_fix, v.fix_seg, v.fix_slot, v.fix_cnt, partial, width, out, ctl, huge);
  }
  if (split) {
    TEMP_LAUNCH(K_FIXUP, k_fixup_split, dim3(1024), dim3(256), 0, st, v.fix_seg, v.fix_slot, v.fix_cnt, partial, width, out, ctl, cap, scratch);
  }
}

// forward / dx aggregation into `out` rows of segments that have edges (others untouched)
static int run_agg(int mode, const TempEdgeView& v, const TempMembers* mb, const float* feat, int ldf, const int32_t* ids, const float* W, int n_rel_rows,
                   const float* nnorm, int d_in, int d_out, int num_bases, float* out, float* partial, hipStream_t st) {
  if (v.n_chunks == 0) return TEMP_OK;
  int S = 0;
  bool fixed = false;
  const int wres = (mode == MODE_FWD) ? d_out : d_in;
  if (fast_shape(d_in, d_out, num_bases, &S)) {
#define TEMP_AGG(SS)                                                                                   \
  if (mode == MODE_FWD) fixed = launch_agg<SS, MODE_FWD>(v, mb, 0, feat, ldf, ids, W, n_rel_rows, nnorm, d_in, out, partial, st); \
  else fixed = launch_agg<SS, MODE_DX>(v, mb, 1, feat, ldf, ids, W, n_rel_rows, nnorm, d_in, out, partial, st);
    if (S == 1) { TEMP_AGG(1) } else if (S == 2) { TEMP_AGG(2) } else { TEMP_AGG(4) }
#undef TEMP_AGG
  } else {
    const int si = d_in / num_bases, so = d_out / num_bases;
    int grid = (v.n_chunks + 3) / 4;
    if (grid > 4096) grid = 4096;
    if (mode == MODE_FWD)
      TEMP_LAUNCH(K_RGCN_AGG_FWD, (k_rgcn_agg_generic<MODE_FWD>), dim3(grid), dim3(256), 0, st, v, feat, ldf, ids, W, nnorm, d_in, d_out, si, so,
                         out, partial);
    else
      TEMP_LAUNCH(K_RGCN_AGG_DX, (k_rgcn_agg_generic<MODE_DX>), dim3(grid), dim3(256), 0, st, v, feat, ldf, ids, W, nnorm, d_in, d_out, si, so,
                         out, partial);
  }
  if (!fixed) launch_fixup(v, partial, wres, out, st);
  return launch_status();
}

template <int S>
static bool launch_dw_tile(const TempEdgeView& v, const TileArgs& t, const float* x, const int32_t* x_ids, const float* dz, const float* nnorm, int D,
                           float* dW, float* partial, hipStream_t st) {
  static const bool granted = tile_grant_lds(k_rgcn_dw_t<S>, TILE_LDS_MAX);
  if (!granted) { (void)hipGetLastError(); return false; }
  const int grid = 8 * ceil_div(t.n_members, 8) * t.n_slices;
  TEMP_LAUNCH(K_RGCN_DW, (k_rgcn_dw_t<S>), dim3(grid), dim3(TILE_THREADS), t.lds_bytes, st, v, t, x, x_ids, dz, nnorm, D, dW, partial);
  g_tile_launches.fetch_add(1, std::memory_order_relaxed);
  return true;
}

template <int S>
static bool launch_dw_hybrid(const TempEdgeView& v, const TileArgs& t, const float* x, const int32_t* x_ids, const float* dz, const float* nnorm, int D,
                             float* dW, float* partial, hipStream_t st) {
  static const bool granted = tile_grant_lds(k_rgcn_dw_h<S>, TILE_LDS_MAX);
  if (!granted) { (void)hipGetLastError(); return false; }
  const int grid = 8 * ceil_div(t.n_members, 8) * t.n_slices;
  TEMP_LAUNCH(K_RGCN_DW, (k_rgcn_dw_h<S>), dim3(grid), dim3(TILE_THREADS), t.lds_bytes, st, v, t, x, x_ids, dz, nnorm, D, dW, partial);
  g_tile_launches.fetch_add(1, std::memory_order_relaxed);
  return true;
}

static int run_dw(const TempEdgeView& v, const TempMembers* mb, const float* x, const int32_t* x_ids, const float* dz, const float* nnorm, int d_in, int d_out, int num_bases,
                  int n_rel_rows, float* dW, float* partial, hipStream_t st) {
  const int si = d_in / num_bases, so = d_out / num_bases;
  const size_t wrow = (size_t)num_bases * si * so;
  if (hipMemsetAsync(dW, 0, (size_t)n_rel_rows * wrow * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
  if (v.n_chunks == 0) return TEMP_OK;
  int S = 0;
  TileArgs ta;
  // (measured at the S-gdelt shape: with TWO row sets in LDS the slices get narrow -- 7-8 of a walker's 16 lanes work -- and the
  // tiled weight-gradient kernel is slower than the L2-gather kernel, 151 against 110-120 us: only TEMP_OPT_RGCN_TILE = 2 takes it)
  if (fast_shape(d_in, d_out, num_bases, &S) && mb && option(TEMP_OPT_RGCN_TILE) >= 3 && n_rel_rows <= 65535 && v.n_partial < 0xffffff &&
      tile_plan(*mb, 2, d_in, S, 0, 0, 4, &ta) &&
      (S == 1 ? launch_dw_hybrid<1>(v, ta, x, x_ids, dz, nnorm, d_in, dW, partial, st)
              : S == 2 ? launch_dw_hybrid<2>(v, ta, x, x_ids, dz, nnorm, d_in, dW, partial, st) : launch_dw_hybrid<4>(v, ta, x, x_ids, dz, nnorm, d_in, dW, partial, st))) {
    // dz rows in LDS, x rows from L2 (rgcn_tile.hpp: k_rgcn_dw_h).  Measured at the S-gdelt shape, alone on the chip: 98 us
    // against 101 for k_rgcn_dw_s (and 106 for d/dh): no gain -- these kernels are bound by the per-block staging + chunk walk
    // of 2.6 rounds of workgroups, not by the L2 gathers -- so only TEMP_OPT_RGCN_TILE = 3 takes it
  } else if (fast_shape(d_in, d_out, num_bases, &S) && mb && option(TEMP_OPT_RGCN_TILE) == 2 && n_rel_rows <= 65535 && v.n_partial < 0xffffff &&
      tile_plan(*mb, 2, d_in, S, 1, 0, 2, &ta) &&
      (S == 1 ? launch_dw_tile<1>(v, ta, x, x_ids, dz, nnorm, d_in, dW, partial, st)
              : S == 2 ? launch_dw_tile<2>(v, ta, x, x_ids, dz, nnorm, d_in, dW, partial, st) : launch_dw_tile<4>(v, ta, x, x_ids, dz, nnorm, d_in, dW, partial, st))) {
    // LDS-tiled (rgcn_tile.hpp)
  } else if (fast_shape(d_in, d_out, num_bases, &S)) {
    const int lpr = pick_lpr(d_in);
    int grid = (v.n_chunks + 3) / 4;
    grid = grid < 8 ? 8 : (grid > 2048 ? 2048 : (grid + 7) / 8 * 8);
    if (lpr == 64 && !rgcn_scalar_off()) {
      if (S == 1) TEMP_LAUNCH(K_RGCN_DW, (k_rgcn_dw_s<1>), dim3(grid), dim3(256), 0, st, v, x, x_ids, dz, nnorm, d_in, dW, partial);
      else if (S == 2) TEMP_LAUNCH(K_RGCN_DW, (k_rgcn_dw_s<2>), dim3(grid), dim3(256), 0, st, v, x, x_ids, dz, nnorm, d_in, dW, partial);
      else TEMP_LAUNCH(K_RGCN_DW, (k_rgcn_dw_s<4>), dim3(grid), dim3(256), 0, st, v, x, x_ids, dz, nnorm, d_in, dW, partial);
    } else if (S == 1) TEMP_LAUNCH(K_RGCN_DW, (k_rgcn_dw<1>), dim3(grid), dim3(256), 0, st, v, x, x_ids, dz, nnorm, d_in, lpr, dW, partial);
    else if (S == 2) TEMP_LAUNCH(K_RGCN_DW, (k_rgcn_dw<2>), dim3(grid), dim3(256), 0, st, v, x, x_ids, dz, nnorm, d_in, lpr, dW, partial);
    else TEMP_LAUNCH(K_RGCN_DW, (k_rgcn_dw<4>), dim3(grid), dim3(256), 0, st, v, x, x_ids, dz, nnorm, d_in, lpr, dW, partial);
  } else {
    int grid = (v.n_chunks + 3) / 4;
    if (grid > 4096) grid = 4096;
    TEMP_LAUNCH(K_RGCN_DW, k_rgcn_dw_generic, dim3(grid), dim3(256), 0, st, v, x, x_ids, dz, nnorm, d_in, d_out, si, so, dW, partial);
  }
  launch_fixup(v, partial, (int)wrow, dW, st);
  return launch_status();
}

// out[row] = act( (in_deg[row] > 0 ? out[row] : 0) + bias + t_loop[ids[row]] ): the self-loop term of a layer whose input
// is a row gather of a table, taken from the table's own product table . W_loop instead of a GEMM over every row.
__global__ void __launch_bounds__(256) k_loop_gather_epi(int n, int d4, const int32_t* __restrict__ ids, const int32_t* __restrict__ in_deg,
                                                         const float4* __restrict__ t_loop, const float4* __restrict__ bias, int act,
                                                         DropSpec drop, float4* __restrict__ out) {
  const size_t total = (size_t)n * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / (unsigned)d4), c = (int)(i - (size_t)row * d4);
    float4 v = in_deg[row] > 0 ? out[i] : zero4();
    float4 lm = t_loop[(size_t)ids[row] * d4 + c];
    if (drop.p > 0.f) lm = drop4(drop, (unsigned)row, (unsigned)c * 4, lm);
    v = add4(v, lm);
    if (bias) v = add4(v, bias[c]);
    if (act == TEMP_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    out[i] = v;
  }
}

// run_dw on the side stream, forked from the caller's stream; joined by SideScope
// `tail` (nullable): more work of the call that depends only on what the fork has seen, issued on the side stream behind run_dw
template <class Tail>
static int dw_forked(SideScope& sc, const TempEdgeView& v, const TempMembers* mb, const float* x, const int32_t* x_ids, const float* dz,
                     const float* nnorm, int d_in, int d_out, int num_bases, int n_rel_rows, float* dW, float* partial, Tail tail) {
  SideStream* ss = sc.ss;
  if (hipEventRecord(ss->fork, sc.st) != hipSuccess) return TEMP_E_LAUNCH;
  if (hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) return TEMP_E_LAUNCH;
  sc.forked = true;                              // from here on the side stream depends on the caller's: it must be joined
  int rc = run_dw(v, mb, x, x_ids, dz, nnorm, d_in, d_out, num_bases, n_rel_rows, dW, partial, ss->s);
  if (!rc) rc = tail(ss->s);
  if (hipEventRecord(ss->join, ss->s) != hipSuccess) return TEMP_E_LAUNCH;
  sc.join_recorded = true;
  return rc;
}
static int dw_forked(SideScope& sc, const TempEdgeView& v, const TempMembers* mb, const float* x, const int32_t* x_ids, const float* dz,
                     const float* nnorm, int d_in, int d_out, int num_bases, int n_rel_rows, float* dW, float* partial) {
  return dw_forked(sc, v, mb, x, x_ids, dz, nnorm, d_in, d_out, num_bases, n_rel_rows, dW, partial, [](hipStream_t) { return (int)TEMP_OK; });
}

struct TableBwdWs {
  float *dz, *dzm, *d_h, *part_dx, *part_dw, *seg_dz;
  void *tn, *cs;
  size_t tn_bytes, cs_bytes, total;
};
static TableBwdWs carve_table_bwd(const TempGraph* g, int n_table, int d_in, int d_out, int num_bases, char* base) {
  TableBwdWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
  const size_t wrow = (size_t)num_bases * (d_in / num_bases) * (d_out / num_bases);
  w.dz = (float*)take((size_t)g->n_nodes * d_out * sizeof(float));
  w.dzm = (float*)take((size_t)g->n_nodes * d_out * sizeof(float));
  w.d_h = (float*)take((size_t)g->n_nodes * d_in * sizeof(float));
  w.part_dx = (float*)take(partial_bytes(g->by_src.n_partial, d_in));
  w.part_dw = (float*)take(partial_bytes(g->by_rel.n_partial, wrow));
  w.seg_dz = (float*)take((size_t)n_table * d_out * sizeof(float));
  w.tn_bytes = gemm_tn_workspace(n_table, d_in, d_out);
  w.tn = take(w.tn_bytes);
  w.cs_bytes = colsum_workspace(g->n_nodes, d_out);
  w.cs = take(w.cs_bytes);
  w.total = off + 256;
  return w;
}

}  // namespace temp

using namespace temp;

extern "C" {

long long temp_tile_launches(void) { return g_tile_launches.load(std::memory_order_relaxed); }
void temp_set_debug_buffer(void* device_ptr, size_t words) { g_debug_buf.store((long long*)device_ptr); g_debug_words.store(device_ptr ? words : 0); }

size_t temp_rgcn_table_fwd_workspace(const TempGraph* g, int n_table, int d_out) {
  if (!g || n_table < 0) return 0;
  return partial_bytes(g->by_dst.n_partial, d_out) + align_up((size_t)n_table * d_out * sizeof(float), 256) + 256;
}

int temp_rgcn_table_fwd(const TempGraph* g, const float* table, const int32_t* ids, int n_table, int d_in, int d_out, int num_bases,
                        int n_rel_rows, const float* weight, const float* loop_w, const float* bias, int act, float* out, void* workspace,
                        size_t workspace_bytes, const TempDropout* drop, void* stream) {
  if (!g || !table || !weight || !loop_w || !out || n_table <= 0 || d_in <= 0 || d_out <= 0 || num_bases <= 0 || n_rel_rows <= 0) return TEMP_E_BADARG;
  if (d_in % num_bases || d_out % num_bases || d_in % 4 || d_out % 4) return TEMP_E_UNSUPPORTED;
  if (g->n_nodes < 0 || !view_ok(g->by_dst) || (g->n_nodes > 0 && (!g->nnorm || !g->in_deg || !ids))) return TEMP_E_BADARG;
  if (act != TEMP_ACT_NONE && act != TEMP_ACT_RELU) return TEMP_E_BADARG;
  if (!workspace || workspace_bytes < temp_rgcn_table_fwd_workspace(g, n_table, d_out)) return TEMP_E_WORKSPACE;
  if (g->n_nodes == 0) return TEMP_OK;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  float* t_loop = (float*)((char*)workspace + partial_bytes(g->by_dst.n_partial, d_out));
  int rc = gemm_add_bias_act(K_GEMM_ISO, n_table, d_out, d_in, table, d_in, nullptr, loop_w, d_out, 0, nullptr, 0, nullptr, nullptr, TEMP_ACT_NONE,
                             t_loop, d_out, st);
  if (rc) return rc;
  rc = run_agg(MODE_FWD, g->by_dst, members_of(g), table, d_in, ids, weight, n_rel_rows, g->nnorm, d_in, d_out, num_bases, out, partial, st);
  if (rc) return rc;
  int grid = ceil_div((long long)g->n_nodes * (d_out / 4), 256);
  if (grid > 4096) grid = 4096;
  TEMP_LAUNCH(K_GEMM_LOOP_FWD, k_loop_gather_epi, dim3(grid), dim3(256), 0, st, g->n_nodes, d_out / 4, ids, g->in_deg, (const float4*)t_loop,
              (const float4*)bias, act, drop_spec(drop), (float4*)out);
  return launch_status();
}

size_t temp_rgcn_table_bwd_workspace(const TempGraph* g, int n_table, int d_in, int d_out, int num_bases) {
  if (!g || num_bases <= 0 || d_in <= 0 || d_out <= 0 || n_table < 0) return 0;
  return carve_table_bwd(g, n_table, d_in, d_out, num_bases, nullptr).total;
}

int temp_rgcn_table_bwd(const TempGraph* g, const float* table, const int32_t* ids, const int32_t* inv_ptr, const int32_t* inv_order, int n_table,
                        const float* out, const float* d_out_grad, int d_in, int d_out, int num_bases, int n_rel_rows, const float* weight,
                        const float* loop_w, int has_bias, int act, float* d_table, float* d_weight, float* d_loop_w, float* d_bias,
                        void* workspace, size_t workspace_bytes, const TempDropout* drop, void* stream) {
  if (!g || !table || !d_out_grad || !weight || !loop_w || !d_table || !d_weight || !d_loop_w || !inv_ptr) return TEMP_E_BADARG;
  if (n_table <= 0 || d_in <= 0 || d_out <= 0 || num_bases <= 0 || n_rel_rows <= 0) return TEMP_E_BADARG;
  if (d_in % num_bases || d_out % num_bases || d_in % 4 || d_out % 4 || d_in > 256 || d_out > 256) return TEMP_E_UNSUPPORTED;
  if (act == TEMP_ACT_RELU && !out) return TEMP_E_BADARG;
  if (has_bias && !d_bias) return TEMP_E_BADARG;
  if (!view_ok(g->by_src) || !view_ok(g->by_rel) || (g->n_nodes > 0 && (!g->nnorm || !g->out_deg || !ids || !inv_order))) return TEMP_E_BADARG;
  if (g->by_rel.n_seg != n_rel_rows) return TEMP_E_BADARG;
  if (!workspace || workspace_bytes < temp_rgcn_table_bwd_workspace(g, n_table, d_in, d_out, num_bases)) return TEMP_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const size_t wrow = (size_t)num_bases * (d_in / num_bases) * (d_out / num_bases);
  if (g->n_nodes == 0) {
    if (hipMemsetAsync(d_table, 0, (size_t)n_table * d_in * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (hipMemsetAsync(d_weight, 0, (size_t)n_rel_rows * wrow * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (hipMemsetAsync(d_loop_w, 0, (size_t)d_in * d_out * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (has_bias && hipMemsetAsync(d_bias, 0, (size_t)d_out * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    return TEMP_OK;
  }
  TableBwdWs w = carve_table_bwd(g, n_table, d_in, d_out, num_bases, (char*)workspace);
  const float* dz = d_out_grad;
  int rc;
  if (act == TEMP_ACT_RELU) {
    rc = relu_bwd((size_t)g->n_nodes * d_out, out, d_out_grad, w.dz, st);
    if (rc) return rc;
    dz = w.dz;
  }
  SideScope side(st);                            // relation-weight gradient beside the rest of the backward (see SideStream)
  SideStream* ss = side.ss;
  if (ss) {
    rc = dw_forked(side, g->by_rel, members_of(g), table, ids, dz, g->nnorm, d_in, d_out, num_bases, n_rel_rows, d_weight, w.part_dw);
    if (rc) return rc;
  }
  // aggregation part of d_h per node row, then everything that is linear in the gathered rows is summed per table row FIRST:
  //   d_table = segsum(out_deg > 0 ? d_h : 0) + segsum(dz) . loop_w^T        d_loop_w = table^T . segsum(dz)
  rc = run_agg(MODE_DX, g->by_src, members_of(g), dz, d_out, nullptr, weight, n_rel_rows, g->nnorm, d_in, d_out, num_bases, w.d_h, w.part_dx, st);
  if (rc) return rc;
  const DropSpec ds = drop_spec(drop);
  const float* dzm = dz;                       // gradient of the (dropped-out) self-loop message
  if (ds.p > 0.f) {
    rc = mask_rows(g->n_nodes, d_out, dz, w.dzm, ds, st);
    if (rc) return rc;
    dzm = w.dzm;
  }
  rc = segment_sum_rows2(n_table, inv_ptr, inv_order, d_in, w.d_h, g->out_deg, d_table, d_out, dzm, w.seg_dz, st, g->n_nodes);
  if (rc) return rc;
  rc = gemm_add_bias_act(K_GEMM_LOOP_DX, n_table, d_in, d_out, w.seg_dz, d_out, nullptr, loop_w, d_out, 1, d_table, d_in, nullptr, nullptr,
                         TEMP_ACT_NONE, d_table, d_in, st);
  if (rc) return rc;
  if (!ss) {
    rc = run_dw(g->by_rel, members_of(g), table, ids, dz, g->nnorm, d_in, d_out, num_bases, n_rel_rows, d_weight, w.part_dw, st);
    if (rc) return rc;
  }
  rc = gemm_tn(n_table, d_in, d_out, table, d_in, w.seg_dz, d_out, d_loop_w, d_out, w.tn, w.tn_bytes, st);
  if (rc) return rc;
  if (has_bias) {
    rc = colsum(g->n_nodes, d_out, dz, d_out, d_bias, w.cs, w.cs_bytes, st);
    if (rc) return rc;
  }
  return side.join();
}

size_t temp_rgcn_fwd_workspace(const TempGraph* g, int d_out) {
  if (!g) return 0;
  return partial_bytes(g->by_dst.n_partial, d_out) + 256;
}

int temp_rgcn_fwd(const TempGraph* g, const float* h, const int32_t* h_ids, int d_in, int d_out, int num_bases, int n_rel_rows,
                  const float* weight, const float* loop_w, const float* bias, int act, float* out, void* workspace,
                  size_t workspace_bytes, const TempDropout* drop, void* stream) {
  if (!g || !h || !weight || !loop_w || !out || d_in <= 0 || d_out <= 0 || num_bases <= 0 || n_rel_rows <= 0) return TEMP_E_BADARG;
  if (d_in % num_bases || d_out % num_bases || d_in % 4 || d_out % 4) return TEMP_E_UNSUPPORTED;
  if (g->n_nodes < 0 || !view_ok(g->by_dst) || (g->n_nodes > 0 && (!g->nnorm || !g->in_deg))) return TEMP_E_BADARG;
  if (act != TEMP_ACT_NONE && act != TEMP_ACT_RELU) return TEMP_E_BADARG;
  if (workspace_bytes < temp_rgcn_fwd_workspace(g, d_out) || (!workspace && g->by_dst.n_partial > 0)) return TEMP_E_WORKSPACE;
  if (g->n_nodes == 0) return TEMP_OK;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  int rc = run_agg(MODE_FWD, g->by_dst, members_of(g), h, d_in, h_ids, weight, n_rel_rows, g->nnorm, d_in, d_out, num_bases, out, partial, st);
  if (rc) return rc;
  // out = act( (in_deg>0 ? out : 0) + bias + h . loop_w )       (MFMA fp32 GEMM, fused epilogue)
  const DropSpec ds = drop_spec(drop);
  return gemm_add_bias_act(K_GEMM_LOOP_FWD, g->n_nodes, d_out, d_in, h, d_in, h_ids, loop_w, d_out, 0, out, d_out, g->in_deg, bias, act, out, d_out, st, &ds);
}

struct BwdWs {
  float* dz;        // [n, d_out]  (only when act == relu)
  float* dzm;       // [n, d_out]  dz masked like the forward self-loop message (only with dropout)
  float* part_dx;   // by_src partial slots [n_partial, d_in]
  float* part_dw;   // by_rel partial slots [n_partial, wrow]
  void* tn;         // gemm_tn workspace
  size_t tn_bytes;
  void* cs;         // colsum workspace
  size_t cs_bytes;
  size_t total;
};
static BwdWs carve_bwd(const TempGraph* g, int d_in, int d_out, int num_bases, char* base) {
  BwdWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
  const size_t wrow = (size_t)num_bases * (d_in / num_bases) * (d_out / num_bases);
  w.dz = (float*)take((size_t)g->n_nodes * d_out * sizeof(float));
  w.dzm = (float*)take((size_t)g->n_nodes * d_out * sizeof(float));
  w.part_dx = (float*)take(partial_bytes(g->by_src.n_partial, d_in));
  w.part_dw = (float*)take(partial_bytes(g->by_rel.n_partial, wrow));
  w.tn_bytes = gemm_tn_workspace(g->n_nodes, d_in, d_out);
  w.tn = take(w.tn_bytes);
  w.cs_bytes = colsum_workspace(g->n_nodes, d_out);
  w.cs = take(w.cs_bytes);
  w.total = off + 256;
  return w;
}

size_t temp_rgcn_bwd_workspace(const TempGraph* g, int d_in, int d_out, int num_bases, int n_rel_rows) {
  (void)n_rel_rows;
  if (!g || num_bases <= 0 || d_in <= 0 || d_out <= 0) return 0;
  return carve_bwd(g, d_in, d_out, num_bases, nullptr).total;
}

int temp_rgcn_bwd(const TempGraph* g, const float* h, const float* out, const float* d_out_grad, int d_in, int d_out, int num_bases,
                  int n_rel_rows, const float* weight, const float* loop_w, int has_bias, int act, float* d_h, float* d_weight,
                  float* d_loop_w, float* d_bias, void* workspace, size_t workspace_bytes, const TempDropout* drop, void* stream) {
  if (!g || !h || !d_out_grad || !weight || !loop_w || !d_h || !d_weight || !d_loop_w) return TEMP_E_BADARG;
  if (d_in <= 0 || d_out <= 0 || num_bases <= 0 || n_rel_rows <= 0) return TEMP_E_BADARG;
  if (d_in % num_bases || d_out % num_bases || d_in % 4 || d_out % 4) return TEMP_E_UNSUPPORTED;
  if (act == TEMP_ACT_RELU && !out) return TEMP_E_BADARG;
  if (has_bias && !d_bias) return TEMP_E_BADARG;
  if (!view_ok(g->by_src) || !view_ok(g->by_rel) || (g->n_nodes > 0 && (!g->nnorm || !g->out_deg))) return TEMP_E_BADARG;
  if (g->by_rel.n_seg != n_rel_rows) return TEMP_E_BADARG;
  if (!workspace || workspace_bytes < temp_rgcn_bwd_workspace(g, d_in, d_out, num_bases, n_rel_rows)) return TEMP_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const size_t wrow = (size_t)num_bases * (d_in / num_bases) * (d_out / num_bases);
  if (g->n_nodes == 0) {
    if (hipMemsetAsync(d_weight, 0, (size_t)n_rel_rows * wrow * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (hipMemsetAsync(d_loop_w, 0, (size_t)d_in * d_out * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (has_bias && hipMemsetAsync(d_bias, 0, (size_t)d_out * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    return TEMP_OK;
  }
  BwdWs w = carve_bwd(g, d_in, d_out, num_bases, (char*)workspace);
  const float* dz = d_out_grad;
  int rc;
  if (act == TEMP_ACT_RELU) {
    rc = relu_bwd((size_t)g->n_nodes * d_out, out, d_out_grad, w.dz, st);
    if (rc) return rc;
    dz = w.dz;
  }
  SideScope side(st);                            // relation-weight gradient beside the rest of the backward (see SideStream)
  SideStream* ss = side.ss;
  const DropSpec ds = drop_spec(drop);
  // without dropout the loop-weight and bias gradients read only h and dz, like the relation-weight gradient: they follow it on the
  // side stream (the relation-weight kernel ends ~80 us before the d/dh aggregation at the S-gdelt shape: the loop-weight product
  // then runs beside the aggregation's tail and the self-loop product instead of behind them)
  const bool tail_beside = ss && !(ds.p > 0.f) && !(option(TEMP_OPT_DEBUG) & 0x20000);       // (TEMP_DEBUG bit 17: A/B, in-stream)
  if (ss) {
    rc = dw_forked(side, g->by_rel, members_of(g), h, nullptr, dz, g->nnorm, d_in, d_out, num_bases, n_rel_rows, d_weight, w.part_dw,
                   [&](hipStream_t s2) {
                     if (!tail_beside) return (int)TEMP_OK;
                     int r2 = gemm_tn(g->n_nodes, d_in, d_out, h, d_in, dz, d_out, d_loop_w, d_out, w.tn, w.tn_bytes, s2);
                     if (!r2 && has_bias) r2 = colsum(g->n_nodes, d_out, dz, d_out, d_bias, w.cs, w.cs_bytes, s2);
                     return r2;
                   });
    if (rc) return rc;
  }
  // d_h (aggregation part) over the by-src view, then d_h = (out_deg>0 ? d_h : 0) + dz . loop_w^T
  rc = run_agg(MODE_DX, g->by_src, members_of(g), dz, d_out, nullptr, weight, n_rel_rows, g->nnorm, d_in, d_out, num_bases, d_h, w.part_dx, st);
  if (rc) return rc;
  const float* dzm = dz;                       // gradient of the (dropped-out) self-loop message
  if (ds.p > 0.f) {
    rc = mask_rows(g->n_nodes, d_out, dz, w.dzm, ds, st);
    if (rc) return rc;
    dzm = w.dzm;
  }
  rc = gemm_add_bias_act(K_GEMM_LOOP_DX, g->n_nodes, d_in, d_out, dzm, d_out, nullptr, loop_w, d_out, 1, d_h, d_in, g->out_deg, nullptr, TEMP_ACT_NONE,
                         d_h, d_in, st);
  if (rc) return rc;
  if (!ss) {
    rc = run_dw(g->by_rel, members_of(g), h, nullptr, dz, g->nnorm, d_in, d_out, num_bases, n_rel_rows, d_weight, w.part_dw, st);
    if (rc) return rc;
  }
  if (!tail_beside) {
    rc = gemm_tn(g->n_nodes, d_in, d_out, h, d_in, dzm, d_out, d_loop_w, d_out, w.tn, w.tn_bytes, st);
    if (rc) return rc;
    if (has_bias) {
      rc = colsum(g->n_nodes, d_out, dz, d_out, d_bias, w.cs, w.cs_bytes, st);
      if (rc) return rc;
    }
  }
  return side.join();
}

// The two halves of temp_rgcn_bwd for a layer that sits INSIDE a recurrence (both layers recurrent, models/RRGCN.py:179-204):
// d_h is needed position by position, the weight / bias gradients are sums over ALL positions -- one pass over the union of the
// positions' graphs afterwards instead of a relation-weight kernel, a fix-up, two reductions, a column sum and three additions
// per position.
int temp_rgcn_bwd_dh(const TempGraph* g, const float* out, const float* d_out_grad, int d_in, int d_out, int num_bases, int n_rel_rows,
                     const float* weight, const float* loop_w, int act, float* d_h, float* dz_out, float* dzm_out, void* workspace,
                     size_t workspace_bytes, const TempDropout* drop, void* stream) {
  if (!g || !d_out_grad || !weight || !loop_w || !d_h) return TEMP_E_BADARG;
  if (d_in <= 0 || d_out <= 0 || num_bases <= 0 || n_rel_rows <= 0) return TEMP_E_BADARG;
  if (d_in % num_bases || d_out % num_bases || d_in % 4 || d_out % 4) return TEMP_E_UNSUPPORTED;
  if (act != TEMP_ACT_NONE && act != TEMP_ACT_RELU) return TEMP_E_BADARG;
  if (act == TEMP_ACT_RELU && (!out || !dz_out)) return TEMP_E_BADARG;       // the masked gradient is an output: the weight pass reads it
  if (!view_ok(g->by_src) || (g->n_nodes > 0 && (!g->nnorm || !g->out_deg))) return TEMP_E_BADARG;
  if (!workspace || workspace_bytes < temp_rgcn_bwd_workspace(g, d_in, d_out, num_bases, n_rel_rows)) return TEMP_E_WORKSPACE;
  if (g->n_nodes == 0) return TEMP_OK;
  hipStream_t st = (hipStream_t)stream;
  BwdWs w = carve_bwd(g, d_in, d_out, num_bases, (char*)workspace);
  const float* dz = d_out_grad;
  int rc;
  if (act == TEMP_ACT_RELU) {
    rc = relu_bwd((size_t)g->n_nodes * d_out, out, d_out_grad, dz_out, st);
    if (rc) return rc;
    dz = dz_out;
  }
  rc = run_agg(MODE_DX, g->by_src, members_of(g), dz, d_out, nullptr, weight, n_rel_rows, g->nnorm, d_in, d_out, num_bases, d_h, w.part_dx, st);
  if (rc) return rc;
  const DropSpec ds = drop_spec(drop);
  const float* dzm = dz;
  if (ds.p > 0.f) {
    if (!dzm_out) return TEMP_E_BADARG;                          // (the weight pass needs the masked gradient of the self-loop message)
    rc = mask_rows(g->n_nodes, d_out, dz, dzm_out, ds, st);
    if (rc) return rc;
    dzm = dzm_out;
  }
  return gemm_add_bias_act(K_GEMM_LOOP_DX, g->n_nodes, d_in, d_out, dzm, d_out, nullptr, loop_w, d_out, 1, d_h, d_in, g->out_deg, nullptr, TEMP_ACT_NONE,
                           d_h, d_in, st);
}

int temp_rgcn_bwd_weights(const TempGraph* g, const float* h, const float* dz, const float* dzm, int d_in, int d_out, int num_bases, int n_rel_rows,
                          int has_bias, float* d_weight, float* d_loop_w, float* d_bias, void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !h || !dz || !d_weight || !d_loop_w) return TEMP_E_BADARG;
  if (d_in <= 0 || d_out <= 0 || num_bases <= 0 || n_rel_rows <= 0) return TEMP_E_BADARG;
  if (d_in % num_bases || d_out % num_bases || d_in % 4 || d_out % 4) return TEMP_E_UNSUPPORTED;
  if (has_bias && !d_bias) return TEMP_E_BADARG;
  if (!view_ok(g->by_rel) || (g->n_nodes > 0 && !g->nnorm)) return TEMP_E_BADARG;
  if (g->by_rel.n_seg != n_rel_rows) return TEMP_E_BADARG;
  if (!workspace || workspace_bytes < temp_rgcn_bwd_workspace(g, d_in, d_out, num_bases, n_rel_rows)) return TEMP_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const size_t wrow = (size_t)num_bases * (d_in / num_bases) * (d_out / num_bases);
  if (g->n_nodes == 0) {
    if (hipMemsetAsync(d_weight, 0, (size_t)n_rel_rows * wrow * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (hipMemsetAsync(d_loop_w, 0, (size_t)d_in * d_out * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (has_bias && hipMemsetAsync(d_bias, 0, (size_t)d_out * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    return TEMP_OK;
  }
  BwdWs w = carve_bwd(g, d_in, d_out, num_bases, (char*)workspace);
  int rc = run_dw(g->by_rel, members_of(g), h, nullptr, dz, g->nnorm, d_in, d_out, num_bases, n_rel_rows, d_weight, w.part_dw, st);
  if (rc) return rc;
  rc = gemm_tn(g->n_nodes, d_in, d_out, h, d_in, dzm ? dzm : dz, d_out, d_loop_w, d_out, w.tn, w.tn_bytes, st);
  if (rc) return rc;
  if (has_bias) rc = colsum(g->n_nodes, d_out, dz, d_out, d_bias, w.cs, w.cs_bytes, st);
  return rc ? rc : launch_status();
}

}  // extern "C"
