// fp32 GEMM on the bf16 matrix pipe:  C[M, N] = epi( A[M,K] . B )  with fp32-equivalent accuracy.
//
// gfx950 issues v_mfma_f32_32x32x16_bf16 at 16x the flop rate of v_mfma_f32_32x32x2_f32 (2.5 PFLOP/s against 157 TFLOP/s
// dense), so an fp32 product is cheaper as SIX bf16 products than as one fp32 product: every fp32 operand is cut by
// truncation into three bf16 pieces  x = h + m + l  (8 + 8 + 8 significant bits: the cut is EXACT, bf16 has the exponent
// range of fp32), and
//     a.b = ah.bh + (ah.bm + am.bh) + (ah.bl + am.bm + al.bh)  +  [am.bl + al.bm + al.bl : dropped, <= 2^-23 |a.b|]
// Each bf16 x bf16 product is exact in the fp32 accumulator, so the result differs from the exact fp32 product by the dropped
// terms only -- the size of ONE fp32 rounding per product (measured: max error against fp64, relative to sum |a||b|, 1.7e-7
// for this kernel against 2.3e-7 for the fp32 MFMA kernel, tools/bx_probe.hip).  6/16 of the fp32 MFMA issue time.
//
// Structure: a block of 4 waves owns 128 rows x (G <= 7 tiles of 32 columns); wave w owns one 32-row panel and all G tiles
// (16 G accumulator registers).  The K loop runs in slabs of 16: the block splits the slab of B (16 x 32G fp32, read with
// coalesced loads in either storage order) into three bf16 planes in LDS, laid out in MFMA fragment order (lane l's 8
// consecutive k of its column are one 16-byte item, items of a tile/plane contiguous: ds_read_b128, conflict-free); every wave
// splits its own A fragment (row li, k = 8 hh .. +7: two float4 straight from global memory, prefetched one slab ahead) in
// registers.  Operands are swapped as in gemm_panel (weights as the A operand) so a lane owns ONE output row and 4
// consecutive columns per accumulator quad: same epilogue contract.  Two blocks per CU; the slab barrier of one is covered
// by the other.  Row tiles are dealt to the XCDs in contiguous eighths for all column groups, so the re-reads of A by the
// other column groups hit that XCD's L2.
#pragma once
#include "gemm_panel.hpp"

namespace temp {

typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int bx_u32x4 __attribute__((ext_vector_type(4)));

// x = h + m + l exactly; returns the three pieces as fp32 bit patterns whose low 16 bits are zero
// (+-inf: r1 = inf - inf = NaN, so a non-finite input gives NaN where an fp32 product gives inf; both are "non-finite" and the
// 9-instruction pair form below has no slot for a guard -- documented in INTEGRATION.md.)
__device__ __forceinline__ void bx_split(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));              // <= 8 significant bits left: the low half is already zero
}
// two consecutive elements -> one dword per plane (element 0 in the low half)
__device__ __forceinline__ void bx_split2(float x0, float x1, unsigned& H, unsigned& M, unsigned& L) {
  unsigned h0, m0, l0, h1, m1, l1;
  bx_split(x0, h0, m0, l0);
  bx_split(x1, h1, m1, l1);
  H = __builtin_amdgcn_perm(h1, h0, 0x07060302u);            // {h1[31:16], h0[31:16]}
  M = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
  L = __builtin_amdgcn_perm(l1, l0, 0x07060302u);
}
__device__ __forceinline__ void bx_split8(const float4 a, const float4 b, bx_u32x4& H, bx_u32x4& M, bx_u32x4& L) {
  unsigned h, m, l;
  bx_split2(a.x, a.y, h, m, l); H[0] = h; M[0] = m; L[0] = l;
  bx_split2(a.z, a.w, h, m, l); H[1] = h; M[1] = m; L[1] = l;
  bx_split2(b.x, b.y, h, m, l); H[2] = h; M[2] = m; L[2] = l;
  bx_split2(b.z, b.w, h, m, l); H[3] = h; M[3] = m; L[3] = l;
}
// Non-finite operands.  A guard in the split (h = +-inf, m = l = 0) does NOT restore IEEE propagation: the six products pair
// the infinite piece with the OTHER operand's m and l pieces, which are exactly 0 for many finite values (any value with <= 8
// significant bits), and inf . 0 = NaN.  So the contract is the one stated in include/temp_amd.h: a non-finite operand makes the
// outputs that depend on it non-finite (NaN where fp32 arithmetic may give +-inf), and nothing else changes
// (tests/test_gpu_parity_r2.py::test_split_operand_gemm_nonfinite_weights_gpu).
__device__ __forceinline__ bx_bf16x8 bx_frag(const bx_u32x4 v) { return __builtin_bit_cast(bx_bf16x8, v); }

// acc += w . a with the six significant products, small terms first (w = weights-side fragment, a = activations-side)
#define BX_MMA(acc, wh, wm, wl, ah, am, al)                                                    \
  do {                                                                                         \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc, 0, 0, 0);                       \
  } while (0)

#define BX_THREADS 256
#define BX_MIN_ROWS 16384                   // below this the fp32 kernels win (measured, tools/bx_probe.hip): too few 128-row blocks

struct BxGeom {
  int N, K, lda, ldb, trans_b;
  int n_groups;                                // column groups of G tiles; the last one starts at tile n_tiles - G (it overlaps its
  int n_tiles, tail_store;                     // predecessor and stores only its last `tail_store` tiles)
  int row_tiles, per_xcd;                      // 128-row tiles of the largest problem, per XCD
};

// VAR is 0 in the library; tools/bx_probe.hip instantiates ablations (bit0: no A loads, bit1: no epilogue, bit2: no B staging,
// bit3: no MFMAs)
template <int G, int TB, class Epi, int VAR = 0>
__global__ void __launch_bounds__(BX_THREADS, 2) k_gemm_bx(PanelBatch<Epi> batch, BxGeom g) {
  constexpr int ITEMS = G * 32 * 2;                         // (column, k-octet) items of a slab
  constexpr int NI = (ITEMS + BX_THREADS - 1) / BX_THREADS;
  __shared__ __attribute__((aligned(16))) bx_u32x4 Bs[2][G * 3 * 64];
  const PanelProblem<Epi>& pb = batch.p[blockIdx.y];
  const int M = pb.M;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int grp = local % g.n_groups, rt_local = local / g.n_groups;
  const int rt = xcd * g.per_xcd + rt_local;
  if (rt_local >= g.per_xcd || rt * 128 >= M) return;        // uniform
  const bool tail = grp == g.n_groups - 1;
  const int t0 = tail ? g.n_tiles - G : grp * G;
  const int t_store = tail ? G - g.tail_store : 0;
  const int n0 = t0 * 32;
  const int N = g.N, K = g.K, ldb = g.ldb;
  const float* __restrict__ A = pb.A;
  const float* __restrict__ B = pb.B;
  const int32_t* __restrict__ a_idx = pb.a_idx;
  const Epi& epi = pb.epi;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int row = rt * 128 + wave * 32 + li;
  long a_src = -1;
  if (row < M) a_src = a_idx ? (long)a_idx[row] : (long)row;
  const bool a_ok = a_src >= 0;
  const float* aptr = A + (size_t)(a_ok ? a_src : 0) * g.lda + 8 * hh;

  f32x16 acc[G];
#pragma unroll
  for (int t = 0; t < G; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- B slab staging: item = (column c of the group, octet o) = 8 consecutive k of one column; a thread owns NI items for the
  // whole K loop.  Threads past the item count redo an earlier item (same data to the same place: no branch).  Columns past N
  // and k past K read a clamped (valid, finite) address: their products land in discarded columns / meet a zeroed A fragment.
  const float* bptr[NI];
  int bdst[NI], boct[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int it = threadIdx.x + i * BX_THREADS;
    it %= ITEMS;
    int c, o;
    if constexpr (TB) { o = it & 1; c = it >> 1; }           // [n][k]: neighbours read neighbouring 32-byte pieces of a row
    else { c = it % (G * 32); o = it / (G * 32); }           // [k][n]: neighbours read neighbouring columns
    const int n = n0 + c < N ? n0 + c : 0;
    bptr[i] = TB ? B + (size_t)n * ldb + 8 * o : B + (size_t)(8 * o) * ldb + n;
    bdst[i] = ((c >> 5) * 3) * 64 + o * 32 + (c & 31);
    boct[i] = 8 * o;
  }
  float4 br[NI][2];
  auto fetch_b = [&](int k0) {
    if constexpr (VAR & 4) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool ok = k0 + boct[i] < K;
      if constexpr (TB) {
        const float* p = bptr[i] + (ok ? k0 : -boct[i]);
        br[i][0] = ld4(p);
        br[i][1] = ld4(p + 4);
      } else {
        const float* p = bptr[i] + (ok ? (long)k0 * ldb : -(long)boct[i] * ldb);
        br[i][0] = make_float4(p[0], p[ldb], p[2 * ldb], p[3 * ldb]);
        br[i][1] = make_float4(p[4 * (long)ldb], p[5 * (long)ldb], p[6 * (long)ldb], p[7 * (long)ldb]);
      }
    }
  };
  // ---- A fragment: k = k0 + 8 hh .. +7 of the lane's row (HBM stream: fetched two slabs ahead, split one slab ahead)
  auto fetch_a = [&](float4 (&a)[2], int k0) {
    const bool ok = a_ok && (k0 + 8 * hh < K);
    const float* p = aptr + (ok ? k0 : -8 * hh);             // !ok: k = 0..7 of a valid row, zeroed at the split
    if constexpr (VAR & 1) { a[0] = make_float4(1.f, 2.f, 3.f, 4.f); a[1] = a[0]; return; }
    a[0] = ld4(p);
    a[1] = ld4(p + 4);
  };
  // The split work of a slab, cut into chunks of one element pair (11 VALU instructions) so that it can be issued in the
  // shadow of the MFMAs (the matrix pipe takes 32 cycles = 8 issue slots per instruction; a wave issues in order):
  //   chunks 0..3: pairs of the next A fragment;  chunks 4 + 4 i + q: pair q of B item i, the last pair followed by the
  //   three LDS stores of the item.
  float4 a1[2], a2[2];                                       // A of slabs s+1, s+2
  bx_u32x4 AH, AM, AL, NH, NM, NL;                           // split A of slabs s, s+1
  bx_u32x4 BH, BM, BL;                                       // B item being split
  constexpr int NCHUNK = 4 + 4 * NI;
  auto pair_of = [&](const float4 (&v)[2], int q, float& x0, float& x1) {
    const float4 f = v[q >> 1];
    x0 = (q & 1) ? f.z : f.x;
    x1 = (q & 1) ? f.w : f.y;
  };
  auto chunk = [&](int c, int buf, bool a_in) {
    float x0, x1;
    unsigned h, m, l;
    if (c < 4) {
      pair_of(a1, c, x0, x1);
      bx_split2(a_in ? x0 : 0.f, a_in ? x1 : 0.f, h, m, l);
      NH[c] = h; NM[c] = m; NL[c] = l;
    } else {
      if constexpr (VAR & 4) return;
      const int i = (c - 4) >> 2, q = (c - 4) & 3;
      pair_of(br[i], q, x0, x1);
      bx_split2(x0, x1, h, m, l);
      BH[q] = h; BM[q] = m; BL[q] = l;
      if (q == 3) {
        bx_u32x4* d = &Bs[buf][bdst[i]];
        d[0] = BH; d[64] = BM; d[128] = BL;
      }
    }
  };

  const int nslabs = (K + 15) >> 4;
  fetch_b(0);
  fetch_a(a1, 0);
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) chunk(c, 0, a_ok && 8 * hh < K);
  AH = NH; AM = NM; AL = NL;
  fetch_a(a1, 16);
  __syncthreads();
  for (int s = 0; s < nslabs; ++s) {
    const int k0 = s * 16;
    fetch_b(k0 + 16);                                        // unconditional (past the end: clamped re-reads)
    fetch_a(a2, k0 + 32);
    const bool a_in = a_ok && (k0 + 16 + 8 * hh < K);
    const bx_bf16x8 ah = bx_frag(AH), am = bx_frag(AM), al = bx_frag(AL);
    const bx_u32x4* bs = &Bs[s & 1][lane];
    // Tiles go through the matrix pipe in PAIRS: the six products of tile t alternate with those of tile t + 1, so that an
    // MFMA never waits for the accumulator of the one issued just before it; the fragments of the next pair are read from
    // LDS behind the first MFMAs of this pair.
    constexpr int NP = (G + 1) / 2;
    bx_u32x4 wf[2][2][3];                                    // [pair parity][tile of the pair][plane]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (u < G) wf[0][u][p] = bs[(u * 3 + p) * 64];
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const int tA = 2 * pr, tB = 2 * pr + 1;
      const bool two = tB < G;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !two) continue;
          const int t = u ? tB : tA;
          const bx_bf16x8 wh = bx_frag(wf[pr & 1][u][0]), wm = bx_frag(wf[pr & 1][u][1]), wl = bx_frag(wf[pr & 1][u][2]);
          if constexpr (!(VAR & 8)) {
            // small terms first
            if (j == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc[t], 0, 0, 0);
            if (j == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc[t], 0, 0, 0);
            if (j == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, acc[t], 0, 0, 0);
            if (j == 3) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, acc[t], 0, 0, 0);
            if (j == 4) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, acc[t], 0, 0, 0);
            if (j == 5) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc[t], 0, 0, 0);
          } else if (j == 0) {
            acc[t][0] += __builtin_bit_cast(float, wf[pr & 1][u][0][0] ^ wf[pr & 1][u][1][1] ^ wf[pr & 1][u][2][2] ^ AH[0] ^ AM[1] ^ AL[2]);
          }
          if (j == 0 && pr + 1 < NP) {                       // the next pair's fragments
            const int tn = 2 * (pr + 1) + u;
            if (tn < G) {
#pragma unroll
              for (int p = 0; p < 3; ++p) wf[(pr + 1) & 1][u][p] = bs[(tn * 3 + p) * 64];
            }
          }
          if ((slot & 1) && (slot >> 1) < NCHUNK) chunk(slot >> 1, (s + 1) & 1, a_in);
          ++slot;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int c = (G * 6) >> 1; c < NCHUNK; ++c) chunk(c, (s + 1) & 1, a_in);   // narrow groups: what did not fit behind the MFMAs
    a1[0] = a2[0]; a1[1] = a2[1];
    AH = NH; AM = NM; AL = NL;
    if constexpr (!(VAR & 16)) __syncthreads();
  }

  const bool row_ok = row < M;
  const typename Epi::RowCtx rc = epi.row_ctx(row_ok ? row : 0);
  // Epilogue in groups of up to four tiles: ALL the epilogue's own loads of a group (the addend of the self-loop layer: 16
  // float4 per lane) are issued before its first store.  `out` may alias the addend, so the compiler cannot hoist a load
  // above a store itself, and tile by tile every tile paid its own memory round trip (cold inputs: up to 7 x ~2 us per
  // block).  A lane reads and writes only its own elements, each read before its write.
  constexpr int EG = 4;
#pragma unroll
  for (int t0e = 0; t0e < G; t0e += EG) {
    float4 pre[EG][4];
    bool ok[EG][4];
#pragma unroll
    for (int u = 0; u < EG; ++u) {
      const int t = t0e + u;
      if (t >= G) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + t * 32 + 8 * q + 4 * hh;
        ok[u][q] = row_ok && col < N && t >= t_store;
        pre[u][q] = epi.pre4(rc, ok[u][q] ? row : 0, ok[u][q] ? col : 0);
      }
    }
#pragma unroll
    for (int u = 0; u < EG; ++u) {
      const int t = t0e + u;
      if (t >= G) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + t * 32 + 8 * q + 4 * hh;
        if constexpr (VAR & 2) { if (acc[t][4 * q] == 12345.678f) epi.fin4(rc, row, col, zero4(), pre[u][q]); continue; }
        if (ok[u][q]) epi.fin4(rc, row, col, make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]), pre[u][q]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Pre-split weights.  k_gemm_bx splits its slab of B in every block: with M / 128 row tiles that is the same work hundreds
// of times, and VALU issue (not the matrix pipe) becomes the limit (measured: 262 VALU instructions per 42 MFMAs).
// k_bx_pack cuts B ONCE into the three bf16 planes, stored in the order the MFMA fragments are read:
//     packed[(s * n_tiles + t) * 3 + p][lane]   (16 bytes: k = 16 s + 8 hh .. +7 of column 32 t + li, plane p; zero past K / N)
// and k_gemm_bxp stages a slab with plain 16-byte copies.  The packed matrix lives in a scratch slot of the library
// (bx_scratch, gemm_kernels.hip: static device memory, one slot per stream in use, nothing allocated at run time).
template <int TB>
__global__ void __launch_bounds__(256) k_bx_pack(int K, int N, int n_tiles, int n_slabs, const float* __restrict__ B, int ldb,
                                                 bx_u32x4* __restrict__ out) {
  const int lane = threadIdx.x & 63, hh = lane >> 5, li = lane & 31;
  const int st = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (st >= n_slabs * n_tiles) return;
  const int s = st / n_tiles, t = st - s * n_tiles;
  const int k = 16 * s + 8 * hh, n = 32 * t + li;
  float4 v0 = zero4(), v1 = zero4();
  if (n < N && k < K) {                                      // K % 8 == 0: the octet is entirely in or out
    if constexpr (TB) {
      const float* p = B + (size_t)n * ldb + k;
      v0 = ld4(p); v1 = ld4(p + 4);
    } else {
      const float* p = B + (size_t)k * ldb + n;
      const size_t l = (size_t)ldb;
      v0 = make_float4(p[0], p[l], p[2 * l], p[3 * l]);
      v1 = make_float4(p[4 * l], p[5 * l], p[6 * l], p[7 * l]);
    }
  }
  bx_u32x4 H, Mi, L;
  bx_split8(v0, v1, H, Mi, L);
  bx_u32x4* d = out + (size_t)st * 192 + lane;
  d[0] = H; d[64] = Mi; d[128] = L;
}

// Several packs in ONE launch (the weights of both directions' input-gate products, both GRUs' W_hh in both operand orders: a
// step packs ten matrices, each launch a few microseconds of latency on an idle chip).
#define BX_PACK_JOBS 8
struct BxPackJob { const float* B; bx_u32x4* out; int K, N, n_tiles, n_slabs, ldb, trans, unit0; };
struct BxPackJobs { BxPackJob j[BX_PACK_JOBS]; int count, total_units; };
static __global__ void __launch_bounds__(256) k_bx_pack_multi(BxPackJobs jobs) {
  const int lane = threadIdx.x & 63, hh = lane >> 5, li = lane & 31;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= jobs.total_units) return;
  int ji = 0;
#pragma unroll
  for (int i = 1; i < BX_PACK_JOBS; ++i)
    if (i < jobs.count && unit >= jobs.j[i].unit0) ji = i;
  const BxPackJob& jb = jobs.j[ji];
  const int st = unit - jb.unit0;
  const int s = st / jb.n_tiles, t = st - s * jb.n_tiles;
  const int k = 16 * s + 8 * hh, n = 32 * t + li;
  float4 v0 = zero4(), v1 = zero4();
  if (n < jb.N && k < jb.K) {                                // K % 8 == 0: the octet is entirely in or out
    if (jb.trans) {
      const float* p = jb.B + (size_t)n * jb.ldb + k;
      v0 = ld4(p); v1 = ld4(p + 4);
    } else {
      const float* p = jb.B + (size_t)k * jb.ldb + n;
      const size_t l = (size_t)jb.ldb;
      v0 = make_float4(p[0], p[l], p[2 * l], p[3 * l]);
      v1 = make_float4(p[4 * l], p[5 * l], p[6 * l], p[7 * l]);
    }
  }
  bx_u32x4 H, Mi, L;
  bx_split8(v0, v1, H, Mi, L);
  bx_u32x4* d = jb.out + (size_t)st * 192 + lane;
  d[0] = H; d[64] = Mi; d[128] = L;
}
inline void bx_pack_jobs_add(BxPackJobs& jobs, const float* B, bx_u32x4* out, int K, int N, int ldb, int trans) {
  BxPackJob& j = jobs.j[jobs.count++];
  j.B = B; j.out = out; j.K = K; j.N = N; j.n_tiles = ceil_div(N, 32); j.n_slabs = ceil_div(K, 16); j.ldb = ldb; j.trans = trans;
  j.unit0 = jobs.total_units;
  jobs.total_units += j.n_tiles * j.n_slabs;
}

typedef float bx_f2 __attribute__((ext_vector_type(2)));
// one element pair of an A fragment: 9 VALU instructions (2 and, packed subtract, 2 and, packed subtract, 3 byte permutes)
__device__ __forceinline__ void bx_split_pair(float x0, float x1, unsigned& H, unsigned& M, unsigned& L) {
  const bx_f2 x = {x0, x1};
  const bx_f2 h = {__uint_as_float(__float_as_uint(x0) & 0xffff0000u), __uint_as_float(__float_as_uint(x1) & 0xffff0000u)};
  const bx_f2 r1 = x - h;
  const bx_f2 m = {__uint_as_float(__float_as_uint(r1[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(r1[1]) & 0xffff0000u)};
  const bx_f2 r2 = r1 - m;
  H = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);       // the upper halves: no mask needed
  M = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
  L = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
}

struct BxPacked { const bx_u32x4* b[PANEL_MAXP]; };           // packed B of every problem of the batch

// VAR is 0 in the library; tools/bx_probe.hip instantiates ablations (bit0: no A loads, bit1: no epilogue, bit2: no B staging,
// bit3: no MFMAs, bit4: no slab barrier)
template <int G, class Epi, int VAR = 0>
__global__ void __launch_bounds__(BX_THREADS, (G <= 4 ? 3 : 2)) k_gemm_bxp(PanelBatch<Epi> batch, BxGeom g, BxPacked packed) {
  constexpr int PIECES = G * 192;                             // 16-byte pieces of a slab of the group
  constexpr int NPC = (PIECES + BX_THREADS - 1) / BX_THREADS;
  __shared__ __attribute__((aligned(16))) bx_u32x4 Bs[2][PIECES];
  const PanelProblem<Epi>& pb = batch.p[blockIdx.y];
  const int M = pb.M;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int grp = local % g.n_groups, rt_local = local / g.n_groups;
  const int rt = xcd * g.per_xcd + rt_local;
  if (rt_local >= g.per_xcd || rt * 128 >= M) return;        // uniform
  const bool tail = grp == g.n_groups - 1;
  const int t0 = tail ? g.n_tiles - G : grp * G;
  const int t_store = tail ? G - g.tail_store : 0;
  const int n0 = t0 * 32;
  const int N = g.N, K = g.K;
  const float* __restrict__ A = pb.A;
  const int32_t* __restrict__ a_idx = pb.a_idx;
  const Epi& epi = pb.epi;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int row = rt * 128 + wave * 32 + li;
  long a_src = -1;
  if (row < M) a_src = a_idx ? (long)a_idx[row] : (long)row;
  // rows past M and gathered "zero rows" (a_idx < 0) compute on row 0; the former are never stored, the latter are zeroed
  // before the epilogue.  k past K meets the zero padding of the packed B.
  const float* aptr = A + (size_t)(a_src >= 0 ? a_src : 0) * g.lda + 8 * hh;
  const int kclamp = K - 8;                                   // last octet that may be read

  f32x16 acc[G];
#pragma unroll
  for (int t = 0; t < G; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- B: slab s of the group = PIECES consecutive 16-byte pieces of the packed matrix
  const bx_u32x4* bsrc = packed.b[blockIdx.y] + (size_t)t0 * 192;
  const size_t slab_stride = (size_t)g.n_tiles * 192;
  int piece[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) piece[i] = (threadIdx.x + i * BX_THREADS) % PIECES;   // the overhang redoes early pieces
  bx_u32x4 br[NPC];
  const int nslabs = (K + 15) >> 4;
  auto fetch_b = [&](int s) {
    if constexpr (VAR & 4) return;
    const bx_u32x4* p = bsrc + (size_t)(s < nslabs ? s : nslabs - 1) * slab_stride;
#pragma unroll
    for (int i = 0; i < NPC; ++i) br[i] = p[piece[i]];
  };
  auto fetch_a = [&](float4 (&a)[2], int k0) {
    const int k = k0 + 8 * hh;
    const float* p = aptr + (k <= kclamp ? k0 : kclamp - 8 * hh);
    if constexpr (VAR & 1) { a[0] = make_float4(1.f, 2.f, 3.f, 4.f); a[1] = a[0]; return; }
    a[0] = ld4(p);
    a[1] = ld4(p + 4);
  };
  float4 a1[2], a2[2];                                       // A of slabs s+1, s+2
  bx_u32x4 AH, AM, AL, NH, NM, NL;                           // split A of slabs s, s+1
  // the non-MFMA work of a slab in chunks that fit an MFMA shadow: chunks 0..3 = element pairs of the next A fragment,
  // chunks 4.. = one LDS store of the next B slab each
  constexpr int NCHUNK = 4 + NPC;
  auto chunk = [&](int c, int buf) {
    if (c < 4) {
      if constexpr (VAR & 64) { NH[c] = __float_as_uint(a1[0].x); NM[c] = NH[c]; NL[c] = NH[c]; return; }
      const float4 f = a1[c >> 1];
      unsigned h, m, l;
      bx_split_pair((c & 1) ? f.z : f.x, (c & 1) ? f.w : f.y, h, m, l);
      NH[c] = h; NM[c] = m; NL[c] = l;
    } else {
      if constexpr (VAR & 4) return;
      Bs[buf][piece[c - 4]] = br[c - 4];
    }
  };

  fetch_b(0);
  fetch_a(a1, 0);
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) chunk(c, 0);
  AH = NH; AM = NM; AL = NL;
  fetch_a(a1, 16);
  __syncthreads();
  for (int s = 0; s < nslabs; ++s) {
    fetch_b(s + 1);                                          // unconditional (past the end: the last slab again)
    fetch_a(a2, s * 16 + 32);
    const bx_bf16x8 ah = bx_frag(AH), am = bx_frag(AM), al = bx_frag(AL);
    const bx_u32x4* bs = &Bs[s & 1][lane];
    // Tiles go through the matrix pipe in PAIRS (the six products of tile t alternate with those of tile t + 1); the
    // fragments of the next pair are read from LDS behind the first MFMAs of this pair; every second MFMA is followed by
    // one chunk of the other work.  The LDS stores of the next slab come last (their loads have had the whole slab to land).
    constexpr int NP = (G + 1) / 2;
    constexpr int NSLOT = G * 3;                             // chunk slots behind the MFMAs
    constexpr int FIRST_B = NSLOT - NPC > 4 ? NSLOT - NPC : 4;   // slot of the first B store
    bx_u32x4 wf[2][2][3];                                    // [pair parity][tile of the pair][plane]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (u < G) { if constexpr (VAR & 32) wf[0][u][p] = AH; else wf[0][u][p] = bs[(u * 3 + p) * 64]; }
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const bool two = 2 * pr + 1 < G;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !two) continue;
          const int t = 2 * pr + u;
          const bx_bf16x8 wh = bx_frag(wf[pr & 1][u][0]), wm = bx_frag(wf[pr & 1][u][1]), wl = bx_frag(wf[pr & 1][u][2]);
          if constexpr (!(VAR & 8)) {
            // small terms first
            if (j == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc[t], 0, 0, 0);
            if (j == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc[t], 0, 0, 0);
            if (j == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, acc[t], 0, 0, 0);
            if (j == 3) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, acc[t], 0, 0, 0);
            if (j == 4) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, acc[t], 0, 0, 0);
            if (j == 5) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc[t], 0, 0, 0);
          } else if (j == 0) {
            acc[t][0] += __builtin_bit_cast(float, wf[pr & 1][u][0][0] ^ wf[pr & 1][u][1][1] ^ wf[pr & 1][u][2][2] ^ AH[0] ^ AM[1] ^ AL[2]);
          }
          if (j == 0 && pr + 1 < NP) {                       // the next pair's fragments
            const int tn = 2 * (pr + 1) + u;
            if (tn < G) {
#pragma unroll
              for (int p = 0; p < 3; ++p) { if constexpr (VAR & 32) wf[(pr + 1) & 1][u][p] = AM; else wf[(pr + 1) & 1][u][p] = bs[(tn * 3 + p) * 64]; }
            }
          }
          if (slot & 1) {
            const int c = slot >> 1;
            if (c < 4) chunk(c, (s + 1) & 1);
            else if (c >= FIRST_B && c - FIRST_B + 4 < NCHUNK) chunk(c - FIRST_B + 4, (s + 1) & 1);
          }
          ++slot;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {                       // narrow groups: what found no slot behind an MFMA
      const bool done = c < 4 ? c < NSLOT : (FIRST_B + c - 4 < NSLOT);
      if (!done) chunk(c, (s + 1) & 1);
    }
    a1[0] = a2[0]; a1[1] = a2[1];
    AH = NH; AM = NM; AL = NL;
    if constexpr (!(VAR & 16)) __syncthreads();
  }

  const bool row_ok = row < M;
  if (a_idx && a_src < 0) {                                   // a gathered zero row
#pragma unroll
    for (int t = 0; t < G; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  }
  const typename Epi::RowCtx rc = epi.row_ctx(row_ok ? row : 0);
  // Epilogue in groups of up to four tiles: ALL the epilogue's own loads of a group (the addend of the self-loop layer: 16
  // float4 per lane) are issued before its first store.  `out` may alias the addend, so the compiler cannot hoist a load
  // above a store itself, and tile by tile every tile paid its own memory round trip (cold inputs: up to 7 x ~2 us per
  // block).  A lane reads and writes only its own elements, each read before its write.
  constexpr int EG = 4;
#pragma unroll
  for (int t0e = 0; t0e < G; t0e += EG) {
    float4 pre[EG][4];
    bool ok[EG][4];
#pragma unroll
    for (int u = 0; u < EG; ++u) {
      const int t = t0e + u;
      if (t >= G) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + t * 32 + 8 * q + 4 * hh;
        ok[u][q] = row_ok && col < N && t >= t_store;
        pre[u][q] = epi.pre4(rc, ok[u][q] ? row : 0, ok[u][q] ? col : 0);
      }
    }
#pragma unroll
    for (int u = 0; u < EG; ++u) {
      const int t = t0e + u;
      if (t >= G) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + t * 32 + 8 * q + 4 * hh;
        if constexpr (VAR & 2) { if (acc[t][4 * q] == 12345.678f) epi.fin4(rc, row, col, zero4(), pre[u][q]); continue; }
        if (ok[u][q]) epi.fin4(rc, row, col, make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]), pre[u][q]);
      }
    }
  }
}

// temp_set_option(TEMP_OPT_MFMA_BF16X3, 0) keeps every product on the fp32 MFMA kernels (A/B runs, bit-comparison against round 1)
inline bool bx_enabled() { return option(TEMP_OPT_MFMA_BF16X3) != 0; }
// ... and among the split kernels, temp_set_option(TEMP_OPT_MFMA_F16X2, 0) keeps the six-product bf16 split where a three-product
// f16 kernel exists (split_f16.hpp)
inline bool hx_enabled() { return bx_enabled() && option(TEMP_OPT_MFMA_F16X2) != 0; }

// Scratch slot for the packed weights of one launch on `st` (gemm_kernels.hip): nullptr when `bytes` exceed a slot or every
// slot belongs to another stream -- the caller then uses the kernel that splits B itself.
bx_u32x4* bx_scratch(hipStream_t st, size_t bytes);
#define BX_SLOT_BYTES (16u << 20)                            // a slot: packed weights (up to BX_PACK_MAX_BYTES) or k-slice partials
#define BX_PACK_MAX_BYTES (3u << 20)
#define BX_SLOTS 8

inline size_t bx_packed_bytes(int N, int K) { return (size_t)ceil_div(K, 16) * ceil_div(N, 32) * 192 * 16; }

// k4_ok: K % 8 == 4 is planned too (only the slab-staged f16 kernel, gemm_hx.hpp, reads A in quads; the caller checks)
inline bool bx_plan(int N, int K, int lda, int ldb, int trans_b, int max_m, long long total_rows, BxGeom* g, int* G, bool k4_ok = false) {
  if ((k4_ok ? K % 4 : K % 8) || lda % 4 || ldb % 4 || N % 4 || K < 16) return false;
  if (total_rows < BX_MIN_ROWS) return false;
  g->N = N; g->K = K; g->lda = lda; g->ldb = ldb; g->trans_b = trans_b;
  g->n_tiles = ceil_div(N, 32);
  g->row_tiles = ceil_div(max_m, 128);
  g->per_xcd = ceil_div(g->row_tiles, 8);
  // column groups of G <= 7 tiles (the last group is shifted left and recomputes tiles its predecessor stores).  512 blocks
  // run at a time (2 per CU); a block costs its G tiles + half a tile for the split of A: take the width with the cheapest
  // estimate of rounds x block cost.
  int best = 1;
  float best_cost = 1e30f;
  const int gmax = g->n_tiles < 7 ? g->n_tiles : 7;
  for (int w = 1; w <= gmax; ++w) {
    const int ng = ceil_div(g->n_tiles, w);
    // (row tiles of ALL problems of the launch: four 7 500-row problems are 236 row tiles, not 59 -- with the largest problem's
    //  count alone the estimate picked one-tile groups, 1 652 blocks that each split their 128 rows of A again)
    const long long all_tiles = total_rows > 0 ? (total_rows + 127) / 128 : g->row_tiles;
    const long long blocks = (all_tiles > g->row_tiles ? all_tiles : (long long)g->row_tiles) * ng;
    const float rounds = blocks <= 512 ? 1.f : (float)blocks / 512.f + 0.5f;       // the dispatcher back-fills: half a round of tail
    const float cost = rounds * (w + 0.5f);
    if (cost < best_cost * 0.999f || (cost <= best_cost * 1.001f && w > best)) { best_cost = cost; best = w; }
  }
  *G = best;
  g->n_groups = ceil_div(g->n_tiles, best);
  g->tail_store = g->n_tiles - (g->n_groups - 1) * best;
  return true;
}

}  // namespace temp
#include "gemm_bxr.hpp"
#include "gemm_hx.hpp"
namespace temp {

// Pack the weight matrices of the batch (problems that share B share the pack) into this stream's scratch slot.
// -> false when no slot is free or the packs do not fit one (the caller then uses the kernel that splits B itself).
template <class Epi>
static inline bool bx_pack_batch(const PanelBatch<Epi>& batch, int count, const BxGeom& g, hipStream_t st, BxPacked* pk) {
  const size_t pbytes = bx_packed_bytes(g.N, g.K);
  int n_distinct = 0, which[PANEL_MAXP];
  for (int i = 0; i < count; ++i) {
    which[i] = -1;
    for (int j = 0; j < i; ++j)
      if (batch.p[j].B == batch.p[i].B) { which[i] = which[j]; break; }
    if (which[i] < 0) which[i] = n_distinct++;
  }
  bx_u32x4* slot = pbytes * n_distinct <= BX_PACK_MAX_BYTES ? bx_scratch(st, pbytes * n_distinct) : nullptr;
  if (!slot) return false;
  const int n_slabs = ceil_div(g.K, 16);
  int done = 0;
  BxPackJobs jobs = {};
  for (int i = 0; i < PANEL_MAXP; ++i) pk->b[i] = slot;
  for (int i = 0; i < count; ++i) {
    bx_u32x4* dst = slot + (size_t)which[i] * (pbytes / 16);
    pk->b[i] = dst;
    if (which[i] < done) continue;
    ++done;
    if (n_distinct > 1 && n_distinct <= BX_PACK_JOBS) { bx_pack_jobs_add(jobs, batch.p[i].B, dst, g.K, g.N, g.ldb, g.trans_b); continue; }
    const dim3 pgrid(ceil_div((long long)n_slabs * g.n_tiles, 4));
    if (g.trans_b) TEMP_LAUNCH(K_BX_PACK, (k_bx_pack<1>), pgrid, dim3(256), 0, st, g.K, g.N, g.n_tiles, n_slabs, batch.p[i].B, g.ldb, dst);
    else TEMP_LAUNCH(K_BX_PACK, (k_bx_pack<0>), pgrid, dim3(256), 0, st, g.K, g.N, g.n_tiles, n_slabs, batch.p[i].B, g.ldb, dst);
  }
  if (jobs.count) TEMP_LAUNCH(K_BX_PACK, k_bx_pack_multi, dim3(ceil_div(jobs.total_units, 4)), dim3(256), 0, st, jobs);   // all distinct weights of the batch in one launch
  return true;
}

template <int G, class Epi>
static inline void launch_bx_g(int kid, const PanelBatch<Epi>& batch, int count, const BxGeom& g, hipStream_t st, const BxPacked* pk) {
  dim3 grid(8 * g.per_xcd * g.n_groups, count);
  if (pk) {
    TEMP_LAUNCH(kid, (k_gemm_bxp<G, Epi>), grid, dim3(BX_THREADS), 0, st, batch, g, *pk);
    return;
  }
  if (g.trans_b) TEMP_LAUNCH(kid, (k_gemm_bx<G, 1, Epi>), grid, dim3(BX_THREADS), 0, st, batch, g);
  else TEMP_LAUNCH(kid, (k_gemm_bx<G, 0, Epi>), grid, dim3(BX_THREADS), 0, st, batch, g);
}

// weights-resident kernel (gemm_bxr.hpp) for short K; temp_set_option(TEMP_OPT_GEMM_RESIDENT, 0): always the slab-staged kernels
// pk == nullptr: no packed weights -- every block splits its own slice of B (gemm_bxr.hpp)
template <class Epi>
static inline bool launch_bxr(int kid, const PanelBatch<Epi>& batch, int count, const BxGeom& g, hipStream_t st, const BxPacked* pkp) {
  if (!option(TEMP_OPT_GEMM_RESIDENT) || g.K > BXR_MAX_SLABS * 16 || g.K < 72) return false;
  // An epilogue with the raw-load interface that does NOT start the accumulators at its addend (the self-loop product with dropout:
  // the mask applies to the product alone) keeps the addend of four tiles in 64 registers of its own beside the operand stages: the
  // kernel spills (551 us for 116 000 x 200 x 200 against 100 without dropout) -- such products take the slab-staged kernels.
  if (EpiRawPre<Epi>::value && !EpiAccInit<Epi>::value && !(option(TEMP_OPT_DEBUG) & 0x200000)) return false;      // (TEMP_DEBUG bit 21: A/B)
  int max_m = 0;
  for (int i = 0; i < count; ++i) max_m = batch.p[i].M > max_m ? batch.p[i].M : max_m;
  BxrGeom rg;
  if (!bxr_plan(g.N, g.K, g.lda, max_m, &rg)) return false;
  rg.ldb = g.ldb; rg.trans_b = g.trans_b;
  BxPacked pk;
  for (int i = 0; i < PANEL_MAXP; ++i) pk.b[i] = pkp ? pkp->b[i] : nullptr;
  static const bool granted = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_bxr<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, BXR_LDS_BYTES) == hipSuccess;
  if (!granted) { (void)hipGetLastError(); return false; }
  const size_t lds = (size_t)rg.n_slabs * BXR_G * 192 * 16 + BXR_BIAS_BYTES;
  TEMP_LAUNCH(kid, (k_gemm_bxr<Epi>), dim3(256, count), dim3(BXR_WAVES * 64), lds, st, batch, rg, pk);
  return true;
}

template <class Epi>
int launch_gemm_bx(int kid, const PanelBatch<Epi>& batch, int count, const BxGeom& g, int G, hipStream_t st) {
  if (launch_hxr(kid, batch, count, g, st)) return launch_status();               // short K, f16 arithmetic: weights resident (gemm_hxr.hpp)
  if (launch_bxr(kid, batch, count, g, st, nullptr)) return launch_status();      // short K: weights resident, split inside the block
  if (hx_supported(batch, count, g)) {                                            // three f16 products instead of six bf16 ones (gemm_hx.hpp)
    const int rc = launch_gemm_hx(kid, batch, count, g, G, st);
    if (rc != TEMP_E_UNSUPPORTED) return rc;
  }
  BxPacked pk;
  const bool packed = bx_pack_batch(batch, count, g, st, &pk);
  const BxPacked* pp = packed ? &pk : nullptr;
  switch (G) {
    case 1: launch_bx_g<1, Epi>(kid, batch, count, g, st, pp); break;
    case 2: launch_bx_g<2, Epi>(kid, batch, count, g, st, pp); break;
    case 3: launch_bx_g<3, Epi>(kid, batch, count, g, st, pp); break;
    case 4: launch_bx_g<4, Epi>(kid, batch, count, g, st, pp); break;
    case 5: launch_bx_g<5, Epi>(kid, batch, count, g, st, pp); break;
    case 6: launch_bx_g<6, Epi>(kid, batch, count, g, st, pp); break;
    default: launch_bx_g<7, Epi>(kid, batch, count, g, st, pp); break;
  }
  return launch_status();
}

}  // namespace temp
