// k_gru_wgrad (gru_wgrad.hpp) on the f16 matrix pipe: three MFMA products of the scaled two-way split (split_f16.hpp) where the
// bf16 kernel issues six, two operand planes in LDS instead of three (90 KB instead of 135 KB at d = 200).
//
//     d_W_ih[3d, d] = [dr dz dn_i]^T . x        d_W_hh[3d, d] = [dr dz dn_h]^T . hdec        (+ bias sums on a column of ones)
//
// The sums run over the node rows, so the power-of-two scales are per COLUMN of each operand:
//   g4      : col_keys[4d] of the GRU, written by the chain backward (k_gru_chain_bwd_hx: integer maxima, order-independent);
//   x       : x_keys[d], the column maxima of the GRU's input rows (k_absmax_keys below, one pass over x, or the caller's);
//   hdec    : |dec . h| <= 1 -> the constant 2^14 (CHX_STATE_SCALE).
// A staging thread owns the same columns in every slab, so its scales are registers; an output element is unscaled by
// 1 / (scale of its g4 column . scale of its x / hdec column) when the slice's partial is stored -- the slices' partials are plain
// fp32 and are summed in slice order as before (k_gru_wgrad_reduce): bit-repeatable.
// Work decomposition, staging, transpose reads: as in gru_wgrad.hpp (same WgArgs / wg_plan / workspace).
#pragma once
#include "gru_wgrad.hpp"
#include "hx_pack.hpp"

namespace temp {

#ifndef CHX_STATE_SCALE
#define CHX_STATE_SCALE 16384.f
#define CHX_STATE_INV (1.f / 16384.f)
#endif

struct WgxKeys { const unsigned* g[WG_MAXG]; const unsigned* x[WG_MAXG]; };      // per GRU: column keys of g4 [4d] and of x [d]

__host__ __device__ constexpr int wgx_a_bytes() { return 8 * 2 * 1024; }
__host__ __device__ constexpr int wgx_b_bytes(int NT) { return 2 * 16 * wg_bs(NT); }
__host__ __device__ constexpr int wgx_buf_bytes(int NT) { return wgx_a_bytes() + 2 * wgx_b_bytes(NT); }
__host__ __device__ constexpr int wgx_lds_bytes(int NT) { return 2 * wgx_buf_bytes(NT); }

// two consecutive elements with their own scales
__device__ __forceinline__ void hx_split_pair2(float x0, float x1, float s0, float s1, unsigned& H, unsigned& L) {
  const hx_f2 y = {x0 * s0, x1 * s1};
  const hx_f16x2 h = __builtin_convertvector(y, hx_f16x2);
  const hx_f2 r = {y[0] - (float)h[0], y[1] - (float)h[1]};
  const hx_f16x2 l = __builtin_convertvector(r, hx_f16x2);
  H = __builtin_bit_cast(unsigned, h);
  L = __builtin_bit_cast(unsigned, l);
}

// The body is compiled four times -- MIXED (the workgroup stages x AND hdec) x ACTIVE (the wave has a tile) -- so that inside a
// slab there is NO branch (gru_wgrad.hpp).
template <int NT, bool MIXED>
__device__ __forceinline__ void wgx_body(const WgArgs& a, const unsigned* __restrict__ gkeys, const unsigned* __restrict__ xkeys, char* wg_lds,
                                         const WgGroup& G, const int mbeg, const int mend_, const WgOut out, const int b, const int prod, const int vt,
                                         const float* __restrict__ bsrc0, const float* __restrict__ bsrc1, const bool b0_is_x) {
  constexpr int BS = wg_bs(NT), ABYTES = wgx_a_bytes(), BBYTES = wgx_b_bytes(NT), BUF = wgx_buf_bytes(NT);
  const int d = a.d, Ka = 3 * d;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hh = lane >> 5, li = lane & 31;
  const int mend = min(G.M, mend_);
  const int nslabs = mend > mbeg ? (mend - mbeg + 15) >> 4 : 0;
  const int full_slabs = mend > mbeg ? (mend - mbeg) >> 4 : 0;

  // ---- staging items of this thread (gru_wgrad.hpp) and their scales
  const int arow = threadIdx.x >> 5, apc = threadIdx.x & 31;
  int acol;                                                      // first column of the piece inside a g4 row
  bool a_real;
  {
    const WgTile tl = wg_tile(a, b, apc >> 2);
    const int vcol = 32 * tl.vt + 8 * (apc & 3);
    acol = (tl.prod == 1 && vcol >= 2 * d) ? vcol + d : vcol;    // d_W_hh skips the dn_i block
    a_real = !(tl.prod < 0 || vcol >= Ka);
    if (!a_real) acol = 0;                                       // no tile / padding of the last tile: any valid column (its outputs are not stored)
  }
  float sa[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sa[i] = hx_scale(gkeys[acol + i]);
  const int a_lds = (((apc >> 2) * 16) + arow) * 64 + (apc & 3) * 16;   // + plane * 8192
  const unsigned a_voff = (unsigned)(arow * 4 * d + acol) * 4u;         // byte offset inside the slab's rows
  const int ppr = d >> 2, pieces = 16 * ppr;                     // pieces per row, per slab
  unsigned b_voff[2];                                            // byte offset inside the slab's rows
  int b_lds[2];                                                  // LDS byte offset (+ plane * 16 * BS, + BBYTES: second operand)
  float sb[2][4];                                                // scales of the first operand's pieces (x: per column; hdec: the constant)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = ((int)threadIdx.x + 512 * i) % pieces;
    const int br = pc / ppr, bq = pc - br * ppr;
    b_voff[i] = (unsigned)(br * d + 4 * bq) * 4u;
    b_lds[i] = ABYTES + br * BS + bq * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) sb[i][e] = b0_is_x ? hx_scale(xkeys[4 * bq + e]) : CHX_STATE_SCALE;
  }

  constexpr int NLOADS = MIXED ? 6 : 4;                          // 0, 1: the A piece; 2, 3: pieces of the first operand; 4, 5: of the second
  constexpr int BSTG = MIXED ? 1 : 2;
  float4 ra_[2][2];
  float4 rb_[BSTG][MIXED ? 4 : 2];                               // [stage][operand * 2 + piece]
  const size_t ld_g4 = (size_t)4 * d;
  auto fetch1 = [&](auto stage_c, int k, int s, auto ragged_c) {
    constexpr int STG = decltype(stage_c)::value;
    constexpr bool RAG = decltype(ragged_c)::value;
    int m0 = mbeg + 16 * s;
    if constexpr (RAG) { const int last = mbeg + 16 * (nslabs - 1); m0 = m0 < last ? m0 : last; }     // slabs past the end: the last one again (never used)
    if (k < 2) {
      wg_gchar* base = wg_uniform(G.g4 + (size_t)m0 * ld_g4);
      unsigned off = a_voff + 16u * (unsigned)k;
      if constexpr (RAG) {                                       // rows past the slice: its last row (zeroed at the LDS write)
        const int over = m0 + arow - (mend - 1);
        if (over > 0) off -= (unsigned)over * (unsigned)(16 * d);
      }
      ra_[STG][k] = wg_ld16(base, off);
    } else {
      constexpr int BS_ = STG % BSTG;
      const int i = (k - 2) & 1;
      wg_gchar* base = wg_uniform((k < 4 ? bsrc0 : bsrc1) + (size_t)m0 * d);
      unsigned off = b_voff[i];
      if constexpr (RAG) {                                       // rows past the slice: its last row
        const int over = m0 + (b_lds[i] - ABYTES) / BS - (mend - 1);
        if (over > 0) off -= (unsigned)over * (unsigned)(4 * d);
      }
      rb_[BS_][k - 2] = wg_ld16(base, off);
    }
  };
  // staging chunk c of slab s (registers -> LDS buffer `buf`): 0, 1 the halves of the A piece (the two 16-byte fragments-to-be are
  // stored with the second), 2.. the x / hdec pieces
  hx_u32x4 AH, AL;
  auto chunk = [&](auto stage_c, int c, int buf, int s, auto ragged_c) {
    constexpr int STG = decltype(stage_c)::value;
    constexpr bool RAG = decltype(ragged_c)::value;
    char* base = wg_lds + buf * BUF;
    unsigned h, l;
    if (c < 2) {
      float4 f = ra_[STG][c];
      if constexpr (RAG) { if (mbeg + 16 * s + arow >= mend) f = zero4(); }                       // rows past the end contribute nothing
      if (!a_real) f = zero4();                                  // (a padding column scaled by another column's key could overflow: keep it 0)
      hx_split_pair2(f.x, f.y, sa[4 * c], sa[4 * c + 1], h, l); AH[2 * c] = h; AL[2 * c] = l;
      hx_split_pair2(f.z, f.w, sa[4 * c + 2], sa[4 * c + 3], h, l); AH[2 * c + 1] = h; AL[2 * c + 1] = l;
      if (c == 1) {
        *reinterpret_cast<hx_u32x4*>(base + a_lds) = AH; *reinterpret_cast<hx_u32x4*>(base + 8192 + a_lds) = AL;
      }
    } else {
      const float4 f = rb_[STG % BSTG][c - 2];
      hx_u32x2 H, L;
      if (c < 4) {
        const int i = (c - 2) & 1;
        hx_split_pair2(f.x, f.y, sb[i][0], sb[i][1], h, l); H[0] = h; L[0] = l;
        hx_split_pair2(f.z, f.w, sb[i][2], sb[i][3], h, l); H[1] = h; L[1] = l;
      } else {                                                   // (mixed workgroup: the second operand is hdec)
        hx_split_pair(f.x, f.y, CHX_STATE_SCALE, h, l); H[0] = h; L[0] = l;
        hx_split_pair(f.z, f.w, CHX_STATE_SCALE, h, l); H[1] = h; L[1] = l;
      }
      char* p = base + b_lds[(c - 2) & 1] + (c >= 4 ? BBYTES : 0);
      *reinterpret_cast<hx_u32x2*>(p) = H; *reinterpret_cast<hx_u32x2*>(p + 16 * BS) = L;
    }
  };
  typedef std::integral_constant<bool, true> rag_t;
  typedef std::integral_constant<bool, false> full_t;
  typedef std::integral_constant<int, 0> st0_t;
  typedef std::integral_constant<int, 1> st1_t;

  // ---- fragment addresses of this lane
  const int gi16 = lane & 15, g16 = lane >> 4;
  const int fr_row = 8 * hh + (gi16 >> 2), fr_col = 16 * (g16 & 1) + 4 * (gi16 & 3);
  const int fa_off = (wave * 16 + fr_row) * 64 + fr_col * 2;                        // + plane * 8192, + 256 for rows + 4
  const int fb_off = ABYTES + ((MIXED && prod == 1) ? BBYTES : 0) + fr_row * BS + fr_col * 2;  // + plane * 16 BS, + 4 BS for rows + 4, + 64 per tile

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (nslabs > 0) {
#pragma unroll
    for (int k = 0; k < NLOADS; ++k) fetch1(st0_t(), k, 0, rag_t());
#pragma unroll
    for (int c = 0; c < NLOADS; ++c) chunk(st0_t(), c, 0, 0, rag_t());
#pragma unroll
    for (int k = 0; k < NLOADS; ++k) fetch1(st1_t(), k, 1, rag_t());      // (past the end: clamped re-reads, never used)
#pragma unroll
    for (int k = 0; k < (BSTG == 2 ? NLOADS : 2); ++k) fetch1(st0_t(), k, 2, rag_t());
  }
  __syncthreads();

  // One slab: register stage CUR holds slab s + 1, stage 1 - CUR slab s + 2 (in flight).  The slab's own MFMAs come first; the
  // staging of slab s + 1 sits behind the last MFMAs, each LDS write followed at once by the load of slab s + 3 into the register
  // it freed.
  auto slab = [&](auto cur_c, int s, auto ragged_c, auto active_c) {
    constexpr bool ACTIVE = decltype(active_c)::value;
    constexpr int NSTEP = 2 * NLOADS;
    auto stage_step = [&](int i) {
      const int k = i >> 1;
      if (i & 1) fetch1(cur_c, k, (k < 2 || BSTG == 2) ? s + 3 : s + 2, ragged_c);      // (into the register chunk k freed)
      else chunk(cur_c, k, (s + 1) & 1, s + 1, ragged_c);
    };
    if constexpr (!ACTIVE) {                                      // a wave without a tile (mixed / left-over workgroups): staging only
#pragma unroll
      for (int i = 0; i < NSTEP; ++i) stage_step(i);
      __syncthreads();
      return;
    }
    const wg_lds_char* lbase = (const wg_lds_char*)(wg_lds + (s & 1) * BUF);
    const wg_lds_char* la = lbase + fa_off;
    const wg_lds_char* lb = lbase + fb_off;
    const hx_f16x8 ah = hx_frag(wg_tr8(la, 0, 256)), al = hx_frag(wg_tr8(la, 8192, 8192 + 256));
    constexpr int NP = (NT + 1) / 2;
    constexpr int NSLOT = NT * 3;
    constexpr int C0 = NSLOT - NSTEP > 1 ? NSLOT - NSTEP : 1;     // first staging slot
    // fragments of x / hdec, a pair of tiles at a time: [pair parity][tile of the pair] per plane; the l plane of the NEXT pair is read
    // where the current pair's l has had its use (product 0); h is live to the end, so the next pair's h has registers of its own
    hx_u32x4 fh[2][2], fl[2][2];
    auto rd = [&](int p, int tile) { return wg_tr8(lb, p * 16 * BS + tile * 64, p * 16 * BS + 4 * BS + tile * 64); };
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
      if (uu < NT) { fh[0][uu] = rd(0, uu); fl[0][uu] = rd(1, uu); }
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const bool two = 2 * pr + 1 < NT;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
          if (uu == 1 && !two) continue;
          const int t = 2 * pr + uu;
          const hx_f16x8 wh = hx_frag(fh[pr & 1][uu]), wl = hx_frag(fl[pr & 1][uu]);
          // operands swapped (x / hdec first): lane (li, hh) ends up with output row 32 vt + li; small terms first
          if (j == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah, acc[t], 0, 0, 0);
          if (j == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, acc[t], 0, 0, 0);
          if (j == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, acc[t], 0, 0, 0);
          const int tn = 2 * (pr + 1) + uu;
          if (pr + 1 < NP && tn < NT) {
            if (j == 0) fh[(pr + 1) & 1][uu] = rd(0, tn);
            if (j == 1) fl[(pr + 1) & 1][uu] = rd(1, tn);
          }
          if (slot >= C0 && slot - C0 < NSTEP) stage_step(slot - C0);
          ++slot;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // (narrow widths: the steps that found no slot behind an MFMA)
#pragma unroll
    for (int i = 0; i < NSTEP; ++i)
      if (C0 + i >= NSLOT) stage_step(i);
    __syncthreads();
  };
  // (slab s writes slab s + 1 from stage (s + 1) & 1: the prologue put slab 1 into stage 1 and slab 2 into stage 0)
  auto loop = [&](auto active_c) {
    int s = 0;
    for (; s + 4 < full_slabs; s += 2) {                         // slabs up to s + 4 entirely inside the slice: no clamps, no zero fill
      slab(st1_t(), s, full_t(), active_c);
      slab(st0_t(), s + 1, full_t(), active_c);
    }
    for (; s < nslabs; s += 2) {
      slab(st1_t(), s, rag_t(), active_c);
      if (s + 1 < nslabs) slab(st0_t(), s + 1, rag_t(), active_c);
    }
  };
  if (prod < 0) { loop(std::integral_constant<bool, false>()); return; }
  loop(std::integral_constant<bool, true>());

  // ---- lane (li, hh) holds output row 32 vt + li; register quad q of tile t holds columns 32 t + 8 q + 4 hh .. + 3
  const int row = 32 * vt + li;
  if (row >= Ka) return;
  const float ig = hx_inv_scale(gkeys[(prod == 1 && row >= 2 * d) ? row + d : row]);
  float* p = out.part + (size_t)prod * out.pstride + (size_t)(row - out.row0) * d;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int c = t * 32 + 8 * qd + 4 * hh;
      if (c < d) {
        float i0 = ig * CHX_STATE_INV, i1 = i0, i2 = i0, i3 = i0;
        if (prod == 0) { i0 = ig * hx_inv_scale(xkeys[c]); i1 = ig * hx_inv_scale(xkeys[c + 1]); i2 = ig * hx_inv_scale(xkeys[c + 2]); i3 = ig * hx_inv_scale(xkeys[c + 3]); }
        st4(p + c, make_float4(acc[t][4 * qd] * i0, acc[t][4 * qd + 1] * i1, acc[t][4 * qd + 2] * i2, acc[t][4 * qd + 3] * i3));
      } else if (c == d) out.bpart[(size_t)prod * out.bstride + (row - out.row0)] = acc[t][4 * qd] * ig;      // (the column of ones carries no scale)
    }
  }
}

template <int NT>
__global__ void __launch_bounds__(WG_THREADS) k_gru_wgrad_hx(WgArgs a, WgxKeys keys) {
  extern __shared__ __attribute__((aligned(16))) char wg_lds[];
  constexpr int BS = wg_bs(NT), ABYTES = wgx_a_bytes(), BBYTES = wgx_b_bytes(NT), BUF = wgx_buf_bytes(NT);
  // ---- which (GRU, slice) pair, which workgroup of the pair
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int pi = q / a.P, b = q - pi * a.P;
  const int u = xcd * a.per_xcd + pi;
  if (pi >= a.per_xcd || u >= a.count * a.S) return;            // uniform
  const int grp = u / a.S, slice = u - grp * a.S;
  const WgGroup G = a.g[grp];
  const unsigned* gkeys = keys.g[grp];
  const unsigned* xkeys = keys.x[grp];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const WgTile tl = wg_tile(a, b, wave);
  const bool mixed = b >= 2 * a.fb && a.mixed;                   // (never with a.tail: P = 2 fb then)
  const int bprod = b < a.fb ? 0 : (b < 2 * a.fb ? 1 : b - 2 * a.fb);      // the product whose fp32 operand a single-operand workgroup stages
  // ---- LDS: zero both buffers once (padding columns are never written again), then the column of ones (bias sums): column d
  // of the h plane of both x / hdec images, every row
  for (int i = threadIdx.x; i < 2 * BUF / 16; i += WG_THREADS) reinterpret_cast<hx_u32x4*>(wg_lds)[i] = hx_u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  if (threadIdx.x < 64) {                                        // (buffer, operand, row)
    const int buf = threadIdx.x >> 5, op = (threadIdx.x >> 4) & 1, row = threadIdx.x & 15;
    *reinterpret_cast<unsigned short*>(wg_lds + buf * BUF + ABYTES + op * BBYTES + row * BS + 2 * a.d) = 0x3c00;   // f16 1.0
  }
  const int Ka = 3 * a.d;
  const int mbeg = slice * a.rows_per_slice;
  const size_t pslot = (size_t)slice * (2 * a.count) + 2 * grp;
  const WgOut out = {a.part + pslot * ((size_t)Ka * a.d), (size_t)Ka * a.d, a.bpart + pslot * Ka, (size_t)Ka, 0};
  if (mixed) wgx_body<NT, true>(a, gkeys, xkeys, wg_lds, G, mbeg, mbeg + a.rows_per_slice, out, b, tl.prod, tl.vt, G.x, G.hdec, true);
  else wgx_body<NT, false>(a, gkeys, xkeys, wg_lds, G, mbeg, mbeg + a.rows_per_slice, out, b, tl.prod, tl.vt, bprod ? G.hdec : G.x, nullptr, bprod == 0);
  if (a.tail) {
    // the left-over tiles of both products (what a mixed workgroup would take) over this workgroup's share of the slice's rows
    __syncthreads();
    const int R0 = 256 * a.fb, Rt = Ka - R0;
    const WgTile t2 = wg_tile(a, 2 * a.fb, wave);
    const int m2 = mbeg + b * a.rows_per_tail;
    const size_t pslot2 = ((size_t)slice * a.P + b) * (2 * a.count) + 2 * grp;
    const WgOut out2 = {a.part2 + pslot2 * ((size_t)Rt * a.d), (size_t)Rt * a.d, a.bpart2 + pslot2 * Rt, (size_t)Rt, R0};
    wgx_body<NT, true>(a, gkeys, xkeys, wg_lds, G, m2, min(m2 + a.rows_per_tail, mbeg + a.rows_per_slice), out2, 2 * a.fb, t2.prod, t2.vt, G.x, G.hdec, true);
  }
}

}  // namespace temp
