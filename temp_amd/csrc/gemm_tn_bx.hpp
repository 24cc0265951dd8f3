// Weight-gradient products on the bf16 matrix pipe:  out[Ka, Nb] = sum_m A[m, ka] * B[m, nb]  with the exact three-way
// operand split of gemm_bx.hpp (fp32-equivalent accuracy, 6/16 of the fp32 MFMA issue time).
//
// The sum runs over the LONG dimension (m = node rows), so both operands have to be turned: an MFMA fragment is 8
// consecutive m of one column.  A block owns 128 rows of Ka x all NT <= 7 column tiles of Nb over one slice of m (same
// tiling as the fp32 kernel k_gemm_tn<NT, 8, 2>: 8 waves = 4 row tiles x 2 column halves, so every SIMD runs NT tiles).
// Per slab of 16 m the block stages (4 + NT) x 32 columns: a thread loads an 8 (m) x 2 (columns) piece with eight 8-byte
// loads (lanes run along the row: coalesced), splits its 16 elements into the three bf16 planes and writes two fragment
// items per plane (16 bytes: the 8 consecutive m of a column) into LDS in the order the MFMAs read them.  Every element of
// A is split once, every element of B once per 128 rows of Ka.  Loads are two slabs ahead of the MFMAs; the split of slab
// s + 1 is issued in the MFMA shadows of slab s (chunks of one element pair = 9 VALU instructions behind every second
// MFMA); one barrier per slab.  Blocks of one m-slice (the Ka / 128 row blocks) share an XCD, so B is read from HBM once.
// Slices go to a workspace and are summed in slice order by k_reduce_slices (deterministic), as for the fp32 kernel.
#pragma once
#include "gemm_bx.hpp"

namespace temp {

#define TNBX_THREADS 512

// VAR (tools/tnbx_probe.hip only): bit0 no global loads, bit1 no epilogue stores, bit2 no split / LDS stores, bit3 no MFMAs
template <int NT, int VAR = 0>
__global__ void __launch_bounds__(TNBX_THREADS) k_gemm_tn_bx(int M, int Ka, int Nb, const float* __restrict__ A, int lda,
                                                             const float* __restrict__ B, int ldb, int rows_per_slice, int kab,
                                                             int n_slices, float* __restrict__ part, float* __restrict__ bias_part,
                                                             unsigned long long* dbg = nullptr) {
  constexpr int NT_LO = (NT + 1) / 2;
  constexpr int ITEMS_A = 64 * 2, ITEMS_B = NT * 16 * 2, ITEMS = ITEMS_A + ITEMS_B;   // (column pair, m-octet)
  constexpr int LDS_ITEMS = (4 + NT) * 192;                   // 16-byte fragment items of a slab: A tiles, then B tiles
  static_assert(ITEMS <= TNBX_THREADS, "one staging item per thread");
  __shared__ __attribute__((aligned(16))) bx_u32x4 Ls[2][LDS_ITEMS];
  // blocks b, b + 8, ... share an XCD: the kab row blocks of an m-slice are neighbours there
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int kb = q % kab, slice = (q / kab) * 8 + xcd;
  if (slice >= n_slices) return;                              // uniform
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: wave-level branches below)
  const int hh = lane >> 5, li = lane & 31;
  const int kt = wave & 3, half = wave >> 2;
  const int ka_blk = kb * 128;
  const int mbeg = slice * rows_per_slice, mend = min(M, mbeg + rows_per_slice);
  const int nslabs = mend > mbeg ? (mend - mbeg + 15) >> 4 : 0;

  // ---- LDS: zero both buffers once (padding columns are never written again); the bias column of ones
  for (int i = threadIdx.x; i < 2 * LDS_ITEMS; i += TNBX_THREADS) {
    bx_u32x4 z = {0u, 0u, 0u, 0u};
    (&Ls[0][0])[i] = z;
  }
  __syncthreads();
  if (bias_part && threadIdx.x < 4) {                          // column Nb (a padding column: Nb % 32 != 0) = 1 for every m
    const int buf = threadIdx.x >> 1, o = threadIdx.x & 1;
    bx_u32x4 one = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    Ls[buf][768 + ((Nb >> 5) * 3) * 64 + o * 32 + (Nb & 31)] = one;
  }

  // ---- staging item of this thread.  Waves 4, 5 hold the 128 items of A, waves 6, 7, 0 and half of 1 the NT x 32 items of B
  // (the overhang of wave 1 redoes early B items: same data, same place), waves 2, 3 none: the operand is uniform per wave.
  const int it_raw = (threadIdx.x + 256) & 511;
  const bool is_a = wave == 4 || wave == 5;                    // scalar
  const int jt = is_a ? it_raw : (it_raw - ITEMS_A) % ITEMS_B;
  const int cpw = is_a ? 64 : NT * 16;                         // column pairs of the operand's block
  const int cp = jt % cpw, oct = jt / cpw;
  const int col = 2 * cp;
  const bool col_ok = is_a ? (ka_blk + col < Ka) : (col < Nb);
  const bool stager = wave >= 4 || wave < 2;                   // waves 2, 3 have no item
  const float* base = is_a ? A + ka_blk : B;                   // scalar: a load is (scalar row base of the slab) + 32-bit lane offset
  const unsigned ld = is_a ? (unsigned)lda : (unsigned)ldb;
  const int dst = (is_a ? 0 : 768) + ((col >> 5) * 3) * 64 + oct * 32 + (col & 31);   // item of column `col`; column col + 1 is dst + 1
  const int full_slabs = (mend - mbeg) >> 4;                   // slabs with all 16 rows inside the slice
  const unsigned voff = (unsigned)(oct * 8) * ld + (unsigned)(col_ok ? col : 0);   // element offset of the item's first row inside a slab

  float2 R0[8], R1[8];                                        // the item of slabs s + 1, s + 2 (ring of two; a ring of three
                                                              // spills: measured slower)
  // row r of the item of slab s (one 8-byte load per lane)
  auto fetch_row = [&](float2 (&R)[8], int s, int r, auto ragged_c) {
    constexpr bool RAG = decltype(ragged_c)::value;
    if constexpr (VAR & 1) { R[r] = make_float2(1.f, 2.f); return; }
    if constexpr (RAG) {                                       // rows past the slice: its last row (zeroed at the split for A)
      const int m = mbeg + s * 16 + oct * 8 + r;
      const int mc = m < mend ? m : mend - 1;
      R[r] = *reinterpret_cast<const float2*>(base + (size_t)(mc < 0 ? 0 : mc) * ld + (col_ok ? col : 0));
    } else {
      const float* sb = base + (size_t)(mbeg + s * 16 + r) * ld;     // scalar row base + one lane offset
      R[r] = *reinterpret_cast<const float2*>(sb + voff);
    }
  };
  auto fetch = [&](float2 (&R)[8], int s, auto ragged_c) {
#pragma unroll
    for (int r = 0; r < 8; ++r) fetch_row(R, s, r, ragged_c);
  };
  bx_u32x4 SH, SM, SL;                                         // the column being split
  // chunk c of the item: c = 0..3 element pairs (rows 2c, 2c+1) of column 0, 4..7 of column 1; after the last pair of a
  // column its three fragment items are stored
  auto chunk = [&](const float2 (&R)[8], int c, int buf, int s, auto ragged_c) {
    constexpr bool RAG = decltype(ragged_c)::value;
    if constexpr (VAR & 4) return;
    const int j = c & 3, cc = c >> 2;
    float x0 = cc ? R[2 * j].y : R[2 * j].x, x1 = cc ? R[2 * j + 1].y : R[2 * j + 1].x;
    if constexpr (RAG) {
      if (is_a) {                                              // rows past the end of the slice contribute nothing
        const int m0 = mbeg + s * 16 + oct * 8 + 2 * j;
        x0 = m0 < mend ? x0 : 0.f;
        x1 = m0 + 1 < mend ? x1 : 0.f;
      }
    }
    unsigned h, m, l;
    bx_split_pair(x0, x1, h, m, l);
    SH[j] = h; SM[j] = m; SL[j] = l;
    if (j == 3 && col_ok) {
      bx_u32x4* d = &Ls[buf][dst + cc];
      d[0] = SH; d[64] = SM; d[128] = SL;
    }
  };
  typedef std::integral_constant<bool, true> rag_t;
  typedef std::integral_constant<bool, false> full_t;

  auto run = [&](auto ntw_c) {
    constexpr int NTW = decltype(ntw_c)::value;
    constexpr int NP = (NTW + 1) / 2;
    const int t_beg = half * NT_LO;
    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (nslabs > 0) {
      fetch(R0, 0, rag_t());
      fetch(R1, 1, rag_t());
      if (stager) {
#pragma unroll
        for (int c = 0; c < 8; ++c) chunk(R0, c, 0, 0, rag_t());
      }
    }
    __syncthreads();
    // one slab: MFMAs of slab s out of buffer s & 1; the item of slab s + 1 (registers Rc) is split into the other buffer in
    // the MFMA shadows; the item of slab s + 2 is fetched into Rf (= the registers consumed one slab ago)
    auto slab = [&](float2 (&Rc)[8], float2 (&Rf)[8], int s, auto ragged_c) {
      // the eight row loads of slab s + 2 are issued one by one behind the MFMAs of the even slots (issued in a burst at the top,
      // by all eight waves at once, they kept the matrix pipe waiting); past the end: clamped re-reads, never used
      const bx_u32x4* ls = &Ls[s & 1][lane];
      const bx_bf16x8 ah = bx_frag(ls[(kt * 3) * 64]), am = bx_frag(ls[(kt * 3 + 1) * 64]), al = bx_frag(ls[(kt * 3 + 2) * 64]);
      const bx_u32x4* bs = ls + 768 + (size_t)t_beg * 192;
      bx_u32x4 wf[2][2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (u < NTW) { if constexpr (VAR & 32) wf[0][u][p] = SH; else wf[0][u][p] = bs[(u * 3 + p) * 64]; }
      __builtin_amdgcn_sched_barrier(0);
      int slot = 0;
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        const bool two = 2 * pr + 1 < NTW;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const int t = 2 * pr + u;
            const bx_bf16x8 wh = bx_frag(wf[pr & 1][u][0]), wm = bx_frag(wf[pr & 1][u][1]), wl = bx_frag(wf[pr & 1][u][2]);
            if constexpr (!(VAR & 8)) {
              // operands swapped (B first): lane (li, hh) ends up with output row ka0 + li; small terms first
              if (j == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc[t], 0, 0, 0);
              if (j == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc[t], 0, 0, 0);
              if (j == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, acc[t], 0, 0, 0);
              if (j == 3) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, acc[t], 0, 0, 0);
              if (j == 4) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, acc[t], 0, 0, 0);
              if (j == 5) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc[t], 0, 0, 0);
            } else if (j == 0) {
              acc[t][0] += __builtin_bit_cast(float, wf[pr & 1][u][0][0] ^ wf[pr & 1][u][1][1] ^ wf[pr & 1][u][2][2]);
            }
            if (j == 0 && pr + 1 < NP) {
              const int tn = 2 * (pr + 1) + u;
              if (tn < NTW) {
#pragma unroll
                for (int p = 0; p < 3; ++p) { if constexpr (VAR & 32) wf[(pr + 1) & 1][u][p] = SM; else wf[(pr + 1) & 1][u][p] = bs[(tn * 3 + p) * 64]; }
              }
            }
            if (!(slot & 1) && (slot >> 1) < 8) fetch_row(Rf, s + 2, slot >> 1, ragged_c);
            if constexpr (!(VAR & 128)) {
              if ((slot & 1) && (slot >> 1) < 8 && stager) chunk(Rc, slot >> 1, (s + 1) & 1, s + 1, ragged_c);
            } else {                                         // late placement: the last 8 MFMAs
              constexpr int FIRST = NTW * 6 - 8;
              if (slot >= FIRST && stager) chunk(Rc, slot - FIRST, (s + 1) & 1, s + 1, ragged_c);
            }
            ++slot;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#pragma unroll
      for (int r = (NTW * 6 + 1) >> 1; r < 8; ++r) fetch_row(Rf, s + 2, r, ragged_c);      // narrow waves: the rest
      if (stager) {
        if constexpr (!(VAR & 128)) {
#pragma unroll
          for (int c = (NTW * 6) >> 1; c < 8; ++c) chunk(Rc, c, (s + 1) & 1, s + 1, ragged_c);    // narrow waves: the rest
        }
      }
      if constexpr (!(VAR & 16)) __syncthreads();
    };
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if constexpr (VAR & 64) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    // two loops with one latch each (a single loop with both bodies made the register allocator ping-pong the accumulators
    // between two register sets: twice the accumulator registers, spills)
    int s = 0;
    for (; s + 3 < full_slabs; s += 2) {                     // slabs s + 1 .. s + 3 (split or fetched in this round) entirely inside
      slab(R1, R0, s, full_t());                             // the slice: no row clamps, no zero fill
      slab(R0, R1, s + 1, full_t());
    }
    for (; s < nslabs; s += 2) {
      slab(R1, R0, s, rag_t());
      if (s + 1 < nslabs) slab(R0, R1, s + 1, rag_t());
    }
    if constexpr (VAR & 64) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = dbg + 4 * (blockIdx.x * 8 + wave);
        o[0] = t1 - clk_t0; o[1] = clk_r0; o[2] = r1; o[3] = (unsigned long long)nslabs;
      }
    }

    // lane (li, hh) holds output row ka0 + li; register quad q of tile t holds columns (t_beg + t)*32 + 8q + 4hh .. +3
    const int row = ka_blk + kt * 32 + li;
    if (row >= Ka) return;
    float* p = part + (size_t)slice * Ka * Nb + (size_t)row * Nb;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c = (t_beg + t) * 32 + 8 * qd + 4 * hh;
        if constexpr (VAR & 2) { if (acc[t][4 * qd] == 12345.678f) st4(p, zero4()); continue; }
        if (c < Nb) st4(p + c, make_float4(acc[t][4 * qd], acc[t][4 * qd + 1], acc[t][4 * qd + 2], acc[t][4 * qd + 3]));
        else if (bias_part && c == Nb) bias_part[(size_t)slice * Ka + row] = acc[t][4 * qd];
      }
    }
  };
  if (half == 0) run(std::integral_constant<int, NT_LO>());
  else run(std::integral_constant<int, NT - NT_LO>());
}

// Wide variant for Ka > 256: a block owns 256 rows of Ka (8 waves x one 32-row tile each, all NT column tiles per wave: 16 NT
// accumulator registers), so an element of B is split once per 256 rows of Ka instead of once per 128 and every wave both
// stages (one 8 x 2 item per thread) and multiplies -- 2.4 instead of 8 VALU instructions per MFMA on the staging waves.
// VAR as above.
template <int NT, int VAR>
__device__ __forceinline__ void tn_bx8_body(int M, int Ka, int Nb, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                            int rows_per_slice, int kab, int n_slices, float* __restrict__ part,
                                            float* __restrict__ bias_part, unsigned long long* dbg, const int block_id,
                                            const size_t pstride, const size_t bstride) {
  // pstride / bstride: floats between two slices' partials (Ka * Nb / Ka for one product; the multi launch interleaves the
  // products of a slice so that ONE reduction sums all of them)
  constexpr int ITEMS_A = 128 * 2, ITEMS_B = NT * 16 * 2, ITEMS = ITEMS_A + ITEMS_B;   // (column pair, m-octet)
  constexpr int LDS_ITEMS = (8 + NT) * 192;                   // 16-byte fragment items of a slab: A tiles, then B tiles
  static_assert(ITEMS <= TNBX_THREADS, "one staging item per thread");
  __shared__ __attribute__((aligned(16))) bx_u32x4 Ls[2][LDS_ITEMS];
  // blocks b, b + 8, ... share an XCD: the kab row blocks of an m-slice are neighbours there
  const int xcd = block_id & 7, q = block_id >> 3;
  const int kb = q % kab, slice = (q / kab) * 8 + xcd;
  if (slice >= n_slices) return;                              // uniform
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: wave-level branches below)
  const int hh = lane >> 5, li = lane & 31;
  const int kt = wave;
  const int ka_blk = kb * 256;
  const int mbeg = slice * rows_per_slice, mend = min(M, mbeg + rows_per_slice);
  const int nslabs = mend > mbeg ? (mend - mbeg + 15) >> 4 : 0;

  // ---- LDS: zero both buffers once (padding columns are never written again); the bias column of ones
  for (int i = threadIdx.x; i < 2 * LDS_ITEMS; i += TNBX_THREADS) {
    bx_u32x4 z = {0u, 0u, 0u, 0u};
    (&Ls[0][0])[i] = z;
  }
  __syncthreads();
  if (bias_part && threadIdx.x < 4) {                          // column Nb (a padding column: Nb % 32 != 0) = 1 for every m
    const int buf = threadIdx.x >> 1, o = threadIdx.x & 1;
    bx_u32x4 one = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    Ls[buf][1536 + ((Nb >> 5) * 3) * 64 + o * 32 + (Nb & 31)] = one;
  }

  // ---- staging item of this thread: waves 0..3 hold the 256 items of A, waves 4..7 the NT x 32 items of B (the overhang of
  // wave 7 redoes early B items: same data, same place); every wave stages and multiplies.
  const bool is_a = wave < 4;                                  // scalar
  const int jt = is_a ? (int)threadIdx.x : ((int)threadIdx.x - ITEMS_A) % ITEMS_B;
  const int cpw = is_a ? 128 : NT * 16;                        // column pairs of the operand's block
  const int cp = jt % cpw, oct = jt / cpw;
  const int col = 2 * cp;
  const bool col_ok = is_a ? (ka_blk + col < Ka) : (col < Nb);
  constexpr bool stager = true;
  const float* base = is_a ? A + ka_blk : B;                   // scalar: a load is (scalar row base of the slab) + 32-bit lane offset
  const unsigned ld = is_a ? (unsigned)lda : (unsigned)ldb;
  const int dst = (is_a ? 0 : 1536) + ((col >> 5) * 3) * 64 + oct * 32 + (col & 31);   // item of column `col`; column col + 1 is dst + 1
  const int full_slabs = (mend - mbeg) >> 4;                   // slabs with all 16 rows inside the slice
  const unsigned voff = (unsigned)(oct * 8) * ld + (unsigned)(col_ok ? col : 0);   // element offset of the item's first row inside a slab

  float2 R0[8], R1[8];                                        // the item of slabs s + 1, s + 2 (ring of two; a ring of three
                                                              // spills: measured slower)
  // row r of the item of slab s (one 8-byte load per lane)
  auto fetch_row = [&](float2 (&R)[8], int s, int r, auto ragged_c) {
    constexpr bool RAG = decltype(ragged_c)::value;
    if constexpr (VAR & 1) { R[r] = make_float2(1.f, 2.f); return; }
    if constexpr (RAG) {                                       // rows past the slice: its last row (zeroed at the split for A)
      const int m = mbeg + s * 16 + oct * 8 + r;
      const int mc = m < mend ? m : mend - 1;
      R[r] = *reinterpret_cast<const float2*>(base + (size_t)(mc < 0 ? 0 : mc) * ld + (col_ok ? col : 0));
    } else {
      const float* sb = base + (size_t)(mbeg + s * 16 + r) * ld;     // scalar row base + one lane offset
      R[r] = *reinterpret_cast<const float2*>(sb + voff);
    }
  };
  auto fetch = [&](float2 (&R)[8], int s, auto ragged_c) {
#pragma unroll
    for (int r = 0; r < 8; ++r) fetch_row(R, s, r, ragged_c);
  };
  bx_u32x4 SH, SM, SL;                                         // the column being split
  // chunk c of the item: c = 0..3 element pairs (rows 2c, 2c+1) of column 0, 4..7 of column 1; after the last pair of a
  // column its three fragment items are stored
  auto chunk = [&](const float2 (&R)[8], int c, int buf, int s, auto ragged_c) {
    constexpr bool RAG = decltype(ragged_c)::value;
    if constexpr (VAR & 4) return;
    const int j = c & 3, cc = c >> 2;
    float x0 = cc ? R[2 * j].y : R[2 * j].x, x1 = cc ? R[2 * j + 1].y : R[2 * j + 1].x;
    if constexpr (RAG) {
      if (is_a) {                                              // rows past the end of the slice contribute nothing
        const int m0 = mbeg + s * 16 + oct * 8 + 2 * j;
        x0 = m0 < mend ? x0 : 0.f;
        x1 = m0 + 1 < mend ? x1 : 0.f;
      }
    }
    unsigned h, m, l;
    bx_split_pair(x0, x1, h, m, l);
    SH[j] = h; SM[j] = m; SL[j] = l;
    if (j == 3 && col_ok) {
      bx_u32x4* d = &Ls[buf][dst + cc];
      d[0] = SH; d[64] = SM; d[128] = SL;
    }
  };
  typedef std::integral_constant<bool, true> rag_t;
  typedef std::integral_constant<bool, false> full_t;

  auto run = [&](auto ntw_c) {
    constexpr int NTW = decltype(ntw_c)::value;
    constexpr int NP = (NTW + 1) / 2;
    constexpr int t_beg = 0;
    const bool wave_on = ka_blk + kt * 32 < Ka;              // (the last row block of Ka may use fewer waves)
    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (nslabs > 0) {
      fetch(R0, 0, rag_t());
      fetch(R1, 1, rag_t());
      if (stager) {
#pragma unroll
        for (int c = 0; c < 8; ++c) chunk(R0, c, 0, 0, rag_t());
      }
    }
    __syncthreads();
    // one slab: MFMAs of slab s out of buffer s & 1; the item of slab s + 1 (registers Rc) is split into the other buffer in
    // the MFMA shadows; the item of slab s + 2 is fetched into Rf (= the registers consumed one slab ago)
    auto slab = [&](float2 (&Rc)[8], float2 (&Rf)[8], int s, auto ragged_c) {
      // the eight row loads of slab s + 2 are issued one by one behind the MFMAs of the even slots (issued in a burst at the top,
      // by all eight waves at once, they kept the matrix pipe waiting); past the end: clamped re-reads, never used
      const bx_u32x4* ls = &Ls[s & 1][lane];
      const bx_bf16x8 ah = bx_frag(ls[(kt * 3) * 64]), am = bx_frag(ls[(kt * 3 + 1) * 64]), al = bx_frag(ls[(kt * 3 + 2) * 64]);
      const bx_u32x4* bs = ls + 1536;
      bx_u32x4 wf[2][2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (u < NTW) { if constexpr (VAR & 32) wf[0][u][p] = SH; else wf[0][u][p] = bs[(u * 3 + p) * 64]; }
      __builtin_amdgcn_sched_barrier(0);
      int slot = 0;
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        const bool two = 2 * pr + 1 < NTW;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const int t = 2 * pr + u;
            const bx_bf16x8 wh = bx_frag(wf[pr & 1][u][0]), wm = bx_frag(wf[pr & 1][u][1]), wl = bx_frag(wf[pr & 1][u][2]);
            if (!wave_on) {
            } else if constexpr (!(VAR & 8)) {
              // operands swapped (B first): lane (li, hh) ends up with output row ka0 + li; small terms first
              if (j == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc[t], 0, 0, 0);
              if (j == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc[t], 0, 0, 0);
              if (j == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, acc[t], 0, 0, 0);
              if (j == 3) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, acc[t], 0, 0, 0);
              if (j == 4) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, acc[t], 0, 0, 0);
              if (j == 5) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc[t], 0, 0, 0);
            } else if (j == 0) {
              acc[t][0] += __builtin_bit_cast(float, wf[pr & 1][u][0][0] ^ wf[pr & 1][u][1][1] ^ wf[pr & 1][u][2][2]);
            }
            if (j == 0 && pr + 1 < NP) {
              const int tn = 2 * (pr + 1) + u;
              if (tn < NTW) {
#pragma unroll
                for (int p = 0; p < 3; ++p) { if constexpr (VAR & 32) wf[(pr + 1) & 1][u][p] = SM; else wf[(pr + 1) & 1][u][p] = bs[(tn * 3 + p) * 64]; }
              }
            }
            if (!(slot & 1) && (slot >> 1) < 8) fetch_row(Rf, s + 2, slot >> 1, ragged_c);
            if constexpr (!(VAR & 128)) {
              if ((slot & 1) && (slot >> 1) < 8 && stager) chunk(Rc, slot >> 1, (s + 1) & 1, s + 1, ragged_c);
            } else {                                         // late placement: the last 8 MFMAs
              constexpr int FIRST = NTW * 6 - 8;
              if (slot >= FIRST && stager) chunk(Rc, slot - FIRST, (s + 1) & 1, s + 1, ragged_c);
            }
            ++slot;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#pragma unroll
      for (int r = (NTW * 6 + 1) >> 1; r < 8; ++r) fetch_row(Rf, s + 2, r, ragged_c);      // narrow waves: the rest
      if (stager) {
        if constexpr (!(VAR & 128)) {
#pragma unroll
          for (int c = (NTW * 6) >> 1; c < 8; ++c) chunk(Rc, c, (s + 1) & 1, s + 1, ragged_c);    // narrow waves: the rest
        }
      }
      if constexpr (!(VAR & 16)) __syncthreads();
    };
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if constexpr (VAR & 64) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    // two loops with one latch each (a single loop with both bodies made the register allocator ping-pong the accumulators
    // between two register sets: twice the accumulator registers, spills)
    int s = 0;
    for (; s + 3 < full_slabs; s += 2) {                     // slabs s + 1 .. s + 3 (split or fetched in this round) entirely inside
      slab(R1, R0, s, full_t());                             // the slice: no row clamps, no zero fill
      slab(R0, R1, s + 1, full_t());
    }
    for (; s < nslabs; s += 2) {
      slab(R1, R0, s, rag_t());
      if (s + 1 < nslabs) slab(R0, R1, s + 1, rag_t());
    }
    if constexpr (VAR & 64) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = dbg + 4 * (block_id * 8 + wave);
        o[0] = t1 - clk_t0; o[1] = clk_r0; o[2] = r1; o[3] = (unsigned long long)nslabs;
      }
    }

    // lane (li, hh) holds output row ka0 + li; register quad q of tile t holds columns (t_beg + t)*32 + 8q + 4hh .. +3
    const int row = ka_blk + kt * 32 + li;
    if (row >= Ka) return;
    float* p = part + (size_t)slice * pstride + (size_t)row * Nb;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c = (t_beg + t) * 32 + 8 * qd + 4 * hh;
        if constexpr (VAR & 2) { if (acc[t][4 * qd] == 12345.678f) st4(p, zero4()); continue; }
        if (c < Nb) st4(p + c, make_float4(acc[t][4 * qd], acc[t][4 * qd + 1], acc[t][4 * qd + 2], acc[t][4 * qd + 3]));
        else if (bias_part && c == Nb) bias_part[(size_t)slice * bstride + row] = acc[t][4 * qd];
      }
    }
  };
  run(std::integral_constant<int, NT>());
}

template <int NT, int VAR = 0>
__global__ void __launch_bounds__(TNBX_THREADS) k_gemm_tn_bx8(int M, int Ka, int Nb, const float* __restrict__ A, int lda,
                                                             const float* __restrict__ B, int ldb, int rows_per_slice, int kab,
                                                             int n_slices, float* __restrict__ part, float* __restrict__ bias_part,
                                                             unsigned long long* dbg = nullptr) {
  tn_bx8_body<NT, VAR>(M, Ka, Nb, A, lda, B, ldb, rows_per_slice, kab, n_slices, part, bias_part, dbg, (int)blockIdx.x, (size_t)Ka * Nb, (size_t)Ka);
}

// Several products of one shape class in ONE launch (gemm_kernels.hip: gemm_tn_multi): blocks_per_problem (a multiple of 8, so a
// block's XCD is the same as in a launch of its own) consecutive blocks per problem.
struct TnBxBatch { int M[8]; const float* A[8]; const float* B[8]; float* part[8]; float* bpart[8]; size_t pstride, bstride; };
template <int NT>
__global__ void __launch_bounds__(TNBX_THREADS) k_gemm_tn_bx8_multi(TnBxBatch b, int Ka, int Nb, int lda, int ldb, int rows_per_slice, int kab,
                                                                   int n_slices, int blocks_per_problem) {
  if (blocks_per_problem > 0) {
    const int prob = blockIdx.x / blocks_per_problem;
    tn_bx8_body<NT, 0>(b.M[prob], Ka, Nb, b.A[prob], lda, b.B[prob], ldb, rows_per_slice, kab, n_slices, b.part[prob], b.bpart[prob], nullptr,
                       (int)blockIdx.x - prob * blocks_per_problem, b.pstride, b.bstride);
    return;
  }
  // blocks_per_problem = 0: the (product, slice) pairs of ALL products are dealt to the XCDs, 32 blocks (CUs) each -- with
  // per-product dealing, 4 products x 21 slices put 36 blocks on the 32 CUs of five XCDs and the launch took two rounds.
  // An XCD takes per = 32 / kab whole pairs (their kab row blocks share the pair's B slab in that XCD's L2); the 32 - per * kab
  // blocks it has left take row blocks of further pairs that straddle XCDs (tn_multi_units(): the host's count of pairs).
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int per = 32 / kab, left = 32 - per * kab;
  int unit, kb;
  if (q < per * kab) { unit = xcd * per + q / kab; kb = q % kab; }
  else { const int r = xcd * left + (q - per * kab); unit = 8 * per + r / kab; kb = r % kab; }
  const int prob = unit / n_slices, slice = unit - prob * n_slices;
  if (q >= 32 || prob >= 8 || b.M[prob] <= 0) return;
  tn_bx8_body<NT, 0>(b.M[prob], Ka, Nb, b.A[prob], lda, b.B[prob], ldb, rows_per_slice, kab, n_slices, b.part[prob], b.bpart[prob], nullptr,
                     ((slice >> 3) * kab + kb) * 8 + (slice & 7), b.pstride, b.bstride);
}
// (product, slice) pairs one launch of 256 blocks takes: whole pairs per XCD + the pairs made of the XCDs' left-over blocks
inline int tn_multi_units(int kab) {
  const int per = 32 / kab, left = 32 - per * kab;
  return kab > 32 ? 0 : 8 * per + (8 * left) / kab;
}

// shapes this kernel takes: one column block of 5..7 tiles (the split-2 configuration of tn_cfg), even pairs
inline bool tn_bx_ok(int M, int Ka, int Nb, int lda, int ldb) {
  const int nt = ceil_div(Nb, 32);
  // (the 128-row kernel keeps 32-bit element offsets; the wide one -- Ka > 256 -- forms a 64-bit row base per slab)
  const long long span = (long long)M * (lda > ldb ? lda : ldb);
  (void)span;                                                   // (rows beyond the 32-bit span: gemm_tn cuts M into chunks, tn_bx_chunks)
  return bx_enabled() && nt >= 5 && nt <= 7 && Ka % 4 == 0 && Nb % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && M >= 4096;
}
// The 128-row kernel forms 32-bit element offsets from its operands' bases: a product over more rows than that (the HBM-regime
// window: 15 M rows of 200 floats) runs as `chunks` launches over row ranges, each with its own slices of the workspace; one
// reduction over all of them.  (Until round 6 such a product fell back to the fp32 MFMA kernel: 17 ms against 8.)
inline int tn_bx_chunks(int M, int Ka, int lda, int ldb) {
  if (Ka > 256) return 1;                                       // the wide kernel forms a 64-bit row base per slab
  const long long ld = lda > ldb ? lda : ldb, max_rows = ((1ll << 31) - 1) / ld / 256 * 256;
  return (int)((M + max_rows - 1) / max_rows);
}
// m-slices: the kab row blocks of a slice run on one XCD (32 CUs), so an XCD takes floor(32 / kab) slices at a time
inline int tn_bx_slices(int M, int kab) {
  int per_xcd = 32 / kab;
  if (per_xcd < 1) per_xcd = 1;
  int S = 8 * per_xcd;
  const int max_s = (M + 255) / 256;                          // at least 256 rows per slice
  if (S > max_s) S = max_s;
  return S < 1 ? 1 : S;
}

inline bool tn_bx8_ok(int Ka) { return Ka > 256; }
inline int tn_bx8_slices(int M, int kab8) {
  int per_xcd = 32 / kab8;
  if (per_xcd < 1) per_xcd = 1;
  int S = 8 * per_xcd;
  const int max_s = M < 16384 ? (M + 127) / 128 : (M + 255) / 256;   // at least 256 rows per slice (128 for a small M: 48 blocks leave the chip idle)
  if (S > max_s) S = max_s;
  return S < 1 ? 1 : S;
}

template <int NT>
static inline void launch_tn_bx_nt(int M, int Ka, int Nb, const float* A, int lda, const float* B, int ldb, int rps, int kab, int S,
                                   float* part, float* bpart, hipStream_t st) {
  const int grid = 8 * ceil_div(S, 8) * kab;
  if (tn_bx8_ok(Ka)) {                                         // kab = row blocks of 256 here
    TEMP_LAUNCH(K_GEMM_TN_BX8, (k_gemm_tn_bx8<NT>), dim3(grid), dim3(TNBX_THREADS), 0, st, M, Ka, Nb, A, lda, B, ldb, rps, kab, S, part, bpart);
    return;
  }
  TEMP_LAUNCH(K_GEMM_TN_BX, (k_gemm_tn_bx<NT>), dim3(grid), dim3(TNBX_THREADS), 0, st, M, Ka, Nb, A, lda, B, ldb, rps, kab, S, part, bpart);
}

inline void launch_tn_bx(int M, int Ka, int Nb, const float* A, int lda, const float* B, int ldb, int rps, int kab, int S, float* part,
                         float* bpart, hipStream_t st) {
  const int nt = ceil_div(Nb, 32);
  if (nt == 7) launch_tn_bx_nt<7>(M, Ka, Nb, A, lda, B, ldb, rps, kab, S, part, bpart, st);
  else if (nt == 6) launch_tn_bx_nt<6>(M, Ka, Nb, A, lda, B, ldb, rps, kab, S, part, bpart, st);
  else launch_tn_bx_nt<5>(M, Ka, Nb, A, lda, B, ldb, rps, kab, S, part, bpart, st);
}

}  // namespace temp
