// GRU weight gradients of a window chain from ONE gate-gradient matrix (round 5):
//     d_W_ih[3d, d] = [dr dz dn_i]^T . x        d_b_ih = column sums of [dr dz dn_i]
//     d_W_hh[3d, d] = [dr dz dn_h]^T . hdec     d_b_hh = column sums of [dr dz dn_h]
// (the backward of the GRU step of GRRGCNLayer.forward, models/RRGCN.py:84; the sums run over the n node rows of all window
// positions).  The window-chain backward writes its gate gradients ONCE, as g4 = [n][4d] = [dr | dz | dn_i | dn_h] -- two thirds
// of the old `dgh` repeated `dgi` -- and both products read their columns of it through a column map (d_W_hh skips dn_i).
//
// Arithmetic: six bf16 MFMA products of the exact three-way operand split (gemm_bx.hpp), fp32 accumulators.
//
// What is new against gemm_tn_bx.hpp (which turned and split both fp32 operands in registers, 2.4 VALU instructions per MFMA):
//  * The sums run over the ROWS, so an MFMA fragment is 8 consecutive rows of one column.  The workgroup splits a slab of 16 rows
//    into three bf16 planes and stores them in LDS AS THEY LIE (row-major); the LDS transpose-read of gfx950
//    (ds_read_b64_tr_b16: a 4 x 16 block of 16-bit elements, delivered column-wise) then hands every lane the 8 consecutive rows
//    of its column.  No turn in registers, no fragment-order scatter: 0.9 VALU instructions per MFMA.
//  * Work decomposition.  A wave owns one 32-column tile of a product's gate columns ("A tile": 32 output rows of d_W) and all
//    NT = ceil(d / 32) column tiles of x / hdec (16 NT accumulator registers).  The two products of a GRU have T = ceil(3d / 32)
//    A tiles each, addressed through the column map, so NO tile is padded except each product's last: 2 T = 38 wave-tiles at
//    d = 200 where two 600 x 200 products on 256-row blocks paid for 48.  A workgroup = 8 waves = 8 wave-tiles of one product; what
//    is left over of both products (3 + 3 tiles at d = 200) shares one "mixed" workgroup that stages both x and hdec.  The P
//    workgroups of a (GRU, row slice) pair sit on one XCD, so the slice's rows reach that L2 once.
//  * Bias sums ride on a column of ones in the padding of the last column tile of x / hdec.
// Row slices go to a workspace and are summed in slice order by k_reduce_slices: deterministic.
//
// What bounds it (tools/gru_wgrad_probe.hip, headline shape 2 x 60 000 rows, d = 200): the MFMA stream alone takes 450 k shader
// cycles (422 k is the issue bound), the whole kernel 553 k -- but the shader clock falls from 2.16 GHz (MFMAs alone) to
// 1.75-1.87 GHz when the 1.1 GB of operands stream in beside them: the chip's power limit, not a schedule, sets the last 17 %.
#pragma once
#include "gemm_bx.hpp"

namespace temp {

typedef short wg_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) wg_s4 wg_lds_s4;
typedef __attribute__((address_space(3))) char wg_lds_char;

#define WG_THREADS 512
#define WG_MAXG 4

struct WgGroup { int M; const float* g4; const float* x; const float* hdec; };   // one GRU: rows, gate gradients [M][4d], inputs [M][d], decayed states [M][d]
struct WgArgs {
  WgGroup g[WG_MAXG];
  int count, d, S, rows_per_slice;     // GRUs, width, row slices per GRU, rows per slice (a multiple of 16)
  int T, fb, r, mixed, P;              // A tiles per product, full workgroups per product, left-over tiles, one mixed workgroup?, workgroups per (GRU, slice)
  int per_xcd;                         // (GRU, slice) pairs per XCD
  int tail;                            // 1: no mixed workgroup -- the 2 fb full workgroups of a pair each take 1 / (2 fb) of the slice's rows for the left-over tiles afterwards
  int rows_per_tail;                   // rows of such a share (a multiple of 16)
  float* part; float* bpart;           // partials [S][2 count][3d * d], [S][2 count][3d]
  float* part2; float* bpart2;         // tail partials [S * 2 fb][2 count][Rt * d], [S * 2 fb][2 count][Rt], Rt = 3d - 256 fb (the left-over output rows)
  unsigned long long* dbg;             // (probe builds: clock stamps)
};
// where a workgroup's results go: block of product p starts at part + p * pstride, its first row is output row `row0`
struct WgOut { float* part; size_t pstride; float* bpart; size_t bstride; int row0; };

// a wave-uniform GLOBAL pointer, pinned to scalar registers: the load then takes the `scalar base + 32-bit lane offset` form
// instead of a 64-bit address pair per load and register stage (address space 1 is kept: a generic pointer would make these
// flat loads, which also count against the LDS wait counter)
typedef __attribute__((address_space(1))) const char wg_gchar;
__device__ __forceinline__ wg_gchar* wg_uniform(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (wg_gchar*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float4 wg_ld16(wg_gchar* base, unsigned off) {
  return __builtin_bit_cast(float4, *reinterpret_cast<const __attribute__((address_space(1))) bx_u32x4*>(base + off));
}

// two transpose-reads: rows 8 hh .. + 3 and 8 hh + 4 .. + 7 of the lane's column -> the 8 consecutive k of an MFMA fragment.
// Lane i of a 16-lane group supplies the address of row i / 4, columns 4 (i % 4) .. + 3 of the group's 4 x 16 block (8 bytes,
// 8-byte aligned) and receives column i of it (checked on the hardware by tools/gru_wgrad_probe.hip).
__device__ __forceinline__ bx_u32x4 wg_tr8(const wg_lds_char* p, int off0, int off1) {
  const wg_s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s4*)(p + off0));
  const wg_s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s4*)(p + off1));
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 ua = __builtin_bit_cast(u2, a), ub = __builtin_bit_cast(u2, b);
  bx_u32x4 r = {ua[0], ua[1], ub[0], ub[1]};
  return r;
}

// LDS images of a slab (16 rows), three bf16 planes each:
//   A: 8 tiles x [16 rows][64 bytes]            -- four consecutive rows of a tile are one 256-byte bank row
//   x / hdec: [16 rows][BS bytes], BS = NT tiles of 64 bytes, padded so that four consecutive rows fall into four different
//             quarters of the bank row (conflict-free transpose reads of a 4-row block by 32 lanes)
__host__ __device__ constexpr int wg_bs(int NT) { return NT * 64 + ((NT & 1) ? 0 : 64); }
__host__ __device__ constexpr int wg_a_bytes() { return 8 * 3 * 1024; }
__host__ __device__ constexpr int wg_b_bytes(int NT) { return 3 * 16 * wg_bs(NT); }
__host__ __device__ constexpr int wg_buf_bytes(int NT) { return wg_a_bytes() + 2 * wg_b_bytes(NT); }
__host__ __device__ constexpr int wg_lds_bytes(int NT) { return 2 * wg_buf_bytes(NT); }

// The (product, A tile) of wave / tile slot w of workgroup b of a pair: product -1 = none
struct WgTile { int prod, vt; };
__device__ __forceinline__ WgTile wg_tile(const WgArgs& a, int b, int w) {
  WgTile t = {-1, 0};
  if (b < a.fb) { t.prod = 0; t.vt = 8 * b + w; }
  else if (b < 2 * a.fb) { t.prod = 1; t.vt = 8 * (b - a.fb) + w; }
  else if (a.mixed) {
    if (w < a.r) { t.prod = 0; t.vt = 8 * a.fb + w; }
    else if (w < 2 * a.r) { t.prod = 1; t.vt = 8 * a.fb + w - a.r; }
  } else if (w < a.r) { t.prod = b - 2 * a.fb; t.vt = 8 * a.fb + w; }
  return t;
}

// VAR (tools/gru_wgrad_probe.hip only): bit0 no global loads, bit2 no LDS staging writes, bit3 no MFMAs, bit4 no slab barrier,
// bit6 clock stamps
//
// The body is compiled four times -- MIXED (the workgroup stages x AND hdec) x ACTIVE (the wave has a tile) -- so that inside a
// slab there is NO branch: every thread issues the same loads and LDS writes (threads past the piece count of x / hdec redo an
// early piece: same data to the same place).  A conditional load makes the compiler's wait counts assume the SHORTEST load
// queue, i.e. wait for loads issued one slab ago instead of two.
template <int NT, int VAR, bool MIXED>
__device__ __forceinline__ void wg_body(const WgArgs& a, char* wg_lds, const WgGroup& G, const int mbeg, const int mend_, const WgOut out, const int b,
                                        const int prod, const int vt, const float* __restrict__ bsrc0, const float* __restrict__ bsrc1) {
  constexpr int BS = wg_bs(NT), ABYTES = wg_a_bytes(), BBYTES = wg_b_bytes(NT), BUF = wg_buf_bytes(NT);
  const int d = a.d, Ka = 3 * d;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hh = lane >> 5, li = lane & 31;
  const int mend = min(G.M, mend_);
  const int nslabs = mend > mbeg ? (mend - mbeg + 15) >> 4 : 0;
  const int full_slabs = mend > mbeg ? (mend - mbeg) >> 4 : 0;

  // ---- staging items of this thread.
  // A: the workgroup's 8 tiles side by side are 16 rows x 256 columns; thread t takes (row t / 32, columns 8 (t % 32) .. + 7): two
  // 16-byte loads (a wave-instruction reads whole cache lines of two rows), split into the three planes at the LDS write
  const int arow = threadIdx.x >> 5, apc = threadIdx.x & 31;
  int acol;                                                      // first column of the piece inside a g4 row
  {
    const WgTile tl = wg_tile(a, b, apc >> 2);
    const int vcol = 32 * tl.vt + 8 * (apc & 3);
    acol = (tl.prod == 1 && vcol >= 2 * d) ? vcol + d : vcol;    // d_W_hh skips the dn_i block
    if (tl.prod < 0 || vcol >= Ka) acol = 0;                     // no tile / padding of the last tile: any valid column (its outputs are not stored)
  }
  const int a_lds = (((apc >> 2) * 16) + arow) * 64 + (apc & 3) * 16;   // + plane * 8192
  const unsigned a_voff = (unsigned)(arow * 4 * d + acol) * 4u;         // byte offset inside the slab's rows
  // x / hdec: the 16 rows are 4 d sixteen-byte pieces; thread t takes pieces t and t + 512 (modulo the count)
  const int ppr = d >> 2, pieces = 16 * ppr;                     // pieces per row, per slab
  unsigned b_voff[2];                                            // byte offset inside the slab's rows
  int b_lds[2];                                                  // LDS byte offset (+ plane * 16 * BS, + BBYTES: second operand)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = ((int)threadIdx.x + 512 * i) % pieces;
    const int br = pc / ppr, bq = pc - br * ppr;
    b_voff[i] = (unsigned)(br * d + 4 * bq) * 4u;
    b_lds[i] = ABYTES + br * BS + bq * 8;
  }

  // TWO register stages (slabs s + 1 and s + 2 in flight while slab s is multiplied); a mixed workgroup keeps ONE stage for its
  // two fp32 operands (the second stage does not fit the register file; six of its eight waves multiply, so it has the slack)
  constexpr int NLOADS = MIXED ? 6 : 4;                          // 0, 1: the A piece; 2, 3: pieces of the first operand; 4, 5: of the second
  constexpr int BSTG = MIXED ? 1 : 2;
  float4 ra_[2][2];
  float4 rb_[BSTG][MIXED ? 4 : 2];                               // [stage][operand * 2 + piece]
  const size_t ld_g4 = (size_t)4 * d;
  // load k of slab s into register stage STG: (scalar base of the slab's first row) + (32-bit lane offset, constant per thread).
  // ONE load per call: the slab loop issues them one by one behind MFMAs.
  auto fetch1 = [&](auto stage_c, int k, int s, auto ragged_c) {
    constexpr int STG = decltype(stage_c)::value;
    constexpr bool RAG = decltype(ragged_c)::value;
    int m0 = mbeg + 16 * s;
    if constexpr (RAG) { const int last = mbeg + 16 * (nslabs - 1); m0 = m0 < last ? m0 : last; }     // slabs past the end: the last one again (never used)
    if (k < 2) {
      if constexpr (VAR & 1) { ra_[STG][k] = make_float4(1.f, 2.f, 3.f, 4.f); return; }
      wg_gchar* base = wg_uniform(G.g4 + (size_t)m0 * ld_g4);
      unsigned off = a_voff + 16u * (unsigned)k;
      if constexpr (RAG) {                                       // rows past the slice: its last row (zeroed at the LDS write)
        const int over = m0 + arow - (mend - 1);
        if (over > 0) off -= (unsigned)over * (unsigned)(16 * d);
      }
      ra_[STG][k] = wg_ld16(base, off);
    } else {
      constexpr int BS_ = STG % BSTG;
      if constexpr (VAR & 1) { rb_[BS_][k - 2] = make_float4(1.f, 2.f, 3.f, 4.f); return; }
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
  // staging chunk c of slab s (registers -> LDS buffer `buf`), numbered like the loads: 0, 1 the halves of the A piece (the three
  // 16-byte fragments-to-be are stored with the second), 2.. the x / hdec pieces
  bx_u32x4 AH, AM, AL;
  auto chunk = [&](auto stage_c, int c, int buf, int s, auto ragged_c) {
    constexpr int STG = decltype(stage_c)::value;
    constexpr bool RAG = decltype(ragged_c)::value;
    if constexpr (VAR & 4) return;
    char* base = wg_lds + buf * BUF;
    unsigned h, m, l;
    if (c < 2) {
      float4 f = ra_[STG][c];
      if constexpr (RAG) { if (mbeg + 16 * s + arow >= mend) f = zero4(); }                       // rows past the end contribute nothing
      bx_split_pair(f.x, f.y, h, m, l); AH[2 * c] = h; AM[2 * c] = m; AL[2 * c] = l;
      bx_split_pair(f.z, f.w, h, m, l); AH[2 * c + 1] = h; AM[2 * c + 1] = m; AL[2 * c + 1] = l;
      if (c == 1) {
        *reinterpret_cast<bx_u32x4*>(base + a_lds) = AH; *reinterpret_cast<bx_u32x4*>(base + 8192 + a_lds) = AM; *reinterpret_cast<bx_u32x4*>(base + 16384 + a_lds) = AL;
      }
    } else {
      const float4 f = rb_[STG % BSTG][c - 2];
      typedef unsigned int u2 __attribute__((ext_vector_type(2)));
      u2 H, Mi, L;
      bx_split_pair(f.x, f.y, h, m, l); H[0] = h; Mi[0] = m; L[0] = l;
      bx_split_pair(f.z, f.w, h, m, l); H[1] = h; Mi[1] = m; L[1] = l;
      char* p = base + b_lds[(c - 2) & 1] + (c >= 4 ? BBYTES : 0);
      *reinterpret_cast<u2*>(p) = H; *reinterpret_cast<u2*>(p + 16 * BS) = Mi; *reinterpret_cast<u2*>(p + 32 * BS) = L;
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

  // One slab: register stage CUR holds slab s + 1, stage 1 - CUR slab s + 2 (in flight).  The slab's own MFMAs come FIRST; the
  // staging of slab s + 1 (which waits for loads issued a slab and a half ago) sits behind the LAST MFMAs, each LDS write followed
  // at once by the load of slab s + 3 into the register it freed (measured: 5 % against staging at the head of the slab).
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
      if constexpr (!(VAR & 16)) __syncthreads();
      return;
    }
    const wg_lds_char* lbase = (const wg_lds_char*)(wg_lds + (s & 1) * BUF);
    const wg_lds_char* la = lbase + fa_off;
    const wg_lds_char* lb = lbase + fb_off;
    const bx_bf16x8 ah = bx_frag(wg_tr8(la, 0, 256)), am = bx_frag(wg_tr8(la, 8192, 8192 + 256)), al = bx_frag(wg_tr8(la, 16384, 16384 + 256));
    constexpr int NP = (NT + 1) / 2;
    constexpr int NSLOT = NT * 6;
    constexpr int SPREAD = (VAR & 256) ? 2 : 1;                   // a staging step (up to 18 VALU instructions) behind every SPREAD-th MFMA (probe: every second one measured 7 % slower)
    constexpr int C0 = NSLOT - SPREAD * NSTEP > 1 ? NSLOT - SPREAD * NSTEP : 1;     // first staging slot
    // fragments of x / hdec, a pair of tiles at a time: [pair parity][tile of the pair] per plane.  The planes of the NEXT pair are
    // read where the current pair's plane has had its last use (l: after product 0, m: after product 2; h is live to the end, so
    // the next pair's h has registers of its own): 32 fragment registers live instead of 48.
    bx_u32x4 fh[2][2], fm[2][2], fl[2][2];
    auto rd = [&](int p, int tile) { return wg_tr8(lb, p * 16 * BS + tile * 64, p * 16 * BS + 4 * BS + tile * 64); };
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
      if (uu < NT) { fh[0][uu] = rd(0, uu); fm[0][uu] = rd(1, uu); fl[0][uu] = rd(2, uu); }
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const bool two = 2 * pr + 1 < NT;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
          if (uu == 1 && !two) continue;
          const int t = 2 * pr + uu;
          const bx_bf16x8 wh = bx_frag(fh[pr & 1][uu]), wm = bx_frag(fm[pr & 1][uu]), wl = bx_frag(fl[pr & 1][uu]);
          if constexpr (!(VAR & 8)) {
            // operands swapped (x / hdec first): lane (li, hh) ends up with output row 32 vt + li; small terms first
            if (j == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc[t], 0, 0, 0);
            if (j == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, acc[t], 0, 0, 0);
            if (j == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, acc[t], 0, 0, 0);
            if (j == 3) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc[t], 0, 0, 0);
            if (j == 4) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, acc[t], 0, 0, 0);
            if (j == 5) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc[t], 0, 0, 0);
          } else if (j == 0) {
            acc[t][0] += __builtin_bit_cast(float, fh[pr & 1][uu][0] ^ fm[pr & 1][uu][1] ^ fl[pr & 1][uu][2]);
          }
          const int tn = 2 * (pr + 1) + uu;
          if (pr + 1 < NP && tn < NT) {
            if (j == 0) fh[(pr + 1) & 1][uu] = rd(0, tn);
            if (j == 1) fl[(pr + 1) & 1][uu] = rd(2, tn);        // (the current pair's l planes were used by product 0)
            if (j == 3) fm[(pr + 1) & 1][uu] = rd(1, tn);        // (m: products 1 and 2)
          }
          if (slot >= C0 && (slot - C0) % SPREAD == 0 && (slot - C0) / SPREAD < NSTEP) stage_step((slot - C0) / SPREAD);
          ++slot;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // (narrow widths: the steps that found no slot behind an MFMA)
#pragma unroll
    for (int i = 0; i < NSTEP; ++i)
      if (C0 + SPREAD * i >= NSLOT) stage_step(i);
    if constexpr (!(VAR & 16)) __syncthreads();
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
  unsigned long long clk0 = 0, rt0 = 0;
  if constexpr (VAR & 64) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  loop(std::integral_constant<bool, true>());
  if constexpr (VAR & 64) {
    const unsigned long long clk1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && a.dbg) { a.dbg[2 * (blockIdx.x * 8 + wave)] = clk1 - clk0; a.dbg[2 * (blockIdx.x * 8 + wave) + 1] = rt1 - rt0; }
  }

  // ---- lane (li, hh) holds output row 32 vt + li; register quad q of tile t holds columns 32 t + 8 q + 4 hh .. + 3
  const int row = 32 * vt + li;
  if (row >= Ka) return;
  float* p = out.part + (size_t)prod * out.pstride + (size_t)(row - out.row0) * d;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int c = t * 32 + 8 * qd + 4 * hh;
      if (c < d) st4(p + c, make_float4(acc[t][4 * qd], acc[t][4 * qd + 1], acc[t][4 * qd + 2], acc[t][4 * qd + 3]));
      else if (c == d) out.bpart[(size_t)prod * out.bstride + (row - out.row0)] = acc[t][4 * qd];
    }
  }
}

template <int NT, int VAR = 0>
__global__ void __launch_bounds__(WG_THREADS) k_gru_wgrad(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wg_lds[];
  constexpr int BS = wg_bs(NT), ABYTES = wg_a_bytes(), BBYTES = wg_b_bytes(NT), BUF = wg_buf_bytes(NT);
  // ---- which (GRU, slice) pair, which workgroup of the pair
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int pi = q / a.P, b = q - pi * a.P;
  const int u = xcd * a.per_xcd + pi;
  if (pi >= a.per_xcd || u >= a.count * a.S) return;            // uniform
  const int grp = u / a.S, slice = u - grp * a.S;
  const WgGroup G = a.g[grp];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const WgTile tl = wg_tile(a, b, wave);
  const bool mixed = b >= 2 * a.fb && a.mixed;                   // (never with a.tail: P = 2 fb then)
  const int bprod = b < a.fb ? 0 : (b < 2 * a.fb ? 1 : b - 2 * a.fb);      // the product whose fp32 operand a single-operand workgroup stages
  // ---- LDS: zero both buffers once (padding columns are never written again), then the column of ones (bias sums): column d
  // of the h plane of both x / hdec images, every row
  for (int i = threadIdx.x; i < 2 * BUF / 16; i += WG_THREADS) reinterpret_cast<bx_u32x4*>(wg_lds)[i] = bx_u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  if (threadIdx.x < 64) {                                        // (buffer, operand, row)
    const int buf = threadIdx.x >> 5, op = (threadIdx.x >> 4) & 1, row = threadIdx.x & 15;
    *reinterpret_cast<unsigned short*>(wg_lds + buf * BUF + ABYTES + op * BBYTES + row * BS + 2 * a.d) = 0x3f80;   // bf16 1.0
  }
  const int Ka = 3 * a.d;
  const int mbeg = slice * a.rows_per_slice;
  const size_t pslot = (size_t)slice * (2 * a.count) + 2 * grp;
  const WgOut out = {a.part + pslot * ((size_t)Ka * a.d), (size_t)Ka * a.d, a.bpart + pslot * Ka, (size_t)Ka, 0};
  if (mixed) wg_body<NT, VAR, true>(a, wg_lds, G, mbeg, mbeg + a.rows_per_slice, out, b, tl.prod, tl.vt, G.x, G.hdec);
  else wg_body<NT, VAR, false>(a, wg_lds, G, mbeg, mbeg + a.rows_per_slice, out, b, tl.prod, tl.vt, bprod ? G.hdec : G.x, nullptr);
  if (a.tail) {
    // the left-over tiles of both products (what a mixed workgroup would take) over this workgroup's share of the slice's rows
    __syncthreads();
    const int R0 = 256 * a.fb, Rt = Ka - R0;
    const WgTile t2 = wg_tile(a, 2 * a.fb, wave);
    const int m2 = mbeg + b * a.rows_per_tail;
    const size_t pslot2 = ((size_t)slice * a.P + b) * (2 * a.count) + 2 * grp;
    const WgOut out2 = {a.part2 + pslot2 * ((size_t)Rt * a.d), (size_t)Rt * a.d, a.bpart2 + pslot2 * Rt, (size_t)Rt, R0};
    wg_body<NT, VAR, true>(a, wg_lds, G, m2, min(m2 + a.rows_per_tail, mbeg + a.rows_per_slice), out2, 2 * a.fb, t2.prod, t2.vt, G.x, G.hdec);
  }
}

// out[p][row][c] = sum of the row slices' partials, in slice order (deterministic); rows from R0 on (tail layout) have S2 partials
// of Rt rows each.  One thread per four output columns; biases by the threads of column 0.
__global__ void __launch_bounds__(256) k_gru_wgrad_reduce(int nprod, int Ka, int d, int S, const float* __restrict__ part, const float* __restrict__ bpart,
                                                          int R0, int S2, const float* __restrict__ part2, const float* __restrict__ bpart2,
                                                          float* __restrict__ d_w, float* __restrict__ d_b) {
  const int d4 = d >> 2;
  const size_t total = (size_t)nprod * Ka * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d4);
    const size_t pr = i / d4;
    const int row = (int)(pr % Ka), p = (int)(pr / Ka);
    // (eight partials in flight per lane, added in slice order: one at a time the sum was a chain of 32 memory latencies)
    const bool head = row < R0;
    const int Rt = Ka - R0, n = head ? S : S2;
    const size_t rows_p = head ? (size_t)Ka : (size_t)Rt, r = head ? (size_t)row : (size_t)(row - R0);
    const float* src = (head ? part : part2) + ((size_t)p * rows_p + r) * d + 4 * c;
    const float* bsrc = (head ? bpart : bpart2) + (size_t)p * rows_p + r;
    const size_t sstride = (size_t)nprod * rows_p * d, bstride = (size_t)nprod * rows_p;
    float4 acc = zero4();
    float bacc = 0.f;
    for (int s0 = 0; s0 < n; s0 += 8) {
      float4 v[8];
      float bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = s0 + u < n;
        v[u] = ok ? ld4(src + (size_t)(s0 + u) * sstride) : zero4();
        bv[u] = (ok && c == 0) ? bsrc[(size_t)(s0 + u) * bstride] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc = add4(acc, v[u]); bacc += bv[u]; }
    }
    st4(d_w + ((size_t)p * Ka + row) * d + 4 * c, acc);
    if (c == 0) d_b[(size_t)p * Ka + row] = bacc;
  }
}

// host side: the decomposition of `count` GRUs of width d, the longest with max_m rows
inline int wg_col_tiles(int d) { return ceil_div(d + 1, 32); }
inline bool wg_plan(int count, int d, int max_m, WgArgs* a) {
  // (the bias sums ride on a column of ones at column d of the x / hdec images: the launch takes wg_col_tiles(d) = ceil((d + 1) / 32)
  //  column tiles, i.e. one more than the data needs when d is a multiple of 32 -- the reference's default width is 128)
  if (count <= 0 || count > WG_MAXG || d % 8 || d >= 256 || max_m <= 0) return false;
  a->count = count; a->d = d;
  a->T = ceil_div(3 * d, 32); a->fb = a->T / 8; a->r = a->T % 8;
  a->mixed = a->r > 0 && 2 * a->r <= 8;
  // A mixed workgroup stages two fp32 operands out of one register stage and is the launch's straggler (measured: 410 us against
  // 270 for its four neighbours).  With full workgroups around, its work is dealt to them instead: each takes 1 / (2 fb) of the
  // slice's rows for the left-over tiles after its own tiles.
  a->tail = a->mixed && a->fb >= 1;
  a->P = 2 * a->fb + ((a->r && !a->tail) ? (a->mixed ? 1 : 2) : 0);
  if (a->P <= 0 || a->P > 32) return false;
  a->per_xcd = 32 / a->P;
  int S = 8 * a->per_xcd / count;
  const int max_s = (max_m + 255) / 256;                         // at least 256 rows per slice
  if (S > max_s) S = max_s;
  if (S < 1) return false;
  a->S = S;
  const int unit = a->tail ? 16 * a->P : 16;                     // a slice is cut into P shares of whole slabs
  a->rows_per_slice = ceil_div(ceil_div(max_m, S), unit) * unit;
  a->rows_per_tail = a->tail ? a->rows_per_slice / a->P : 0;
  a->dbg = nullptr;
  return true;
}
// workspace: [part | bpart | part2 | bpart2]
struct WgWs { size_t part, bpart, part2, bpart2, xkeys, total; };      // xkeys: [count][d] column keys of the x operands (gru_wgrad_hx.hpp)
inline WgWs wg_workspace(const WgArgs& a) {
  const size_t Ka = 3 * (size_t)a.d, np = 2 * (size_t)a.count;
  const size_t Rt = a.tail ? Ka - 256 * (size_t)a.fb : 0, S2 = a.tail ? (size_t)a.S * a.P : 0;
  WgWs w;
  w.part = 0;
  w.bpart = align_up((size_t)a.S * np * Ka * a.d * sizeof(float), 256);
  w.part2 = w.bpart + align_up((size_t)a.S * np * Ka * sizeof(float), 256);
  w.bpart2 = w.part2 + align_up(S2 * np * Rt * a.d * sizeof(float), 256);
  w.xkeys = w.bpart2 + align_up(S2 * np * Rt * sizeof(float), 256);
  w.total = w.xkeys + align_up((size_t)a.count * (1 + 1024) * a.d * sizeof(unsigned), 256);      // [count][d] keys + [count][ABSMAX_BLOCKS][d] partials (hx_pack.hpp)
  return w;
}

}  // namespace temp
