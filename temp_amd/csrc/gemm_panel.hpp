// Row-panel fp32 MFMA GEMM template shared by gemm_kernels.hip and gru_kernels.hip.
//
//   C[M, n0 .. n0+NT*32) = epi( A[M,K] . B[K, ...] )
//
// A wave owns 32 rows x (NT*32) columns; the 4 waves of a block own 4 consecutive row tiles and
// share the B chunk (KC x NT*32) staged in LDS.  A is read straight from global memory: the MFMA
// sums over k in any order, so lane (row i, half hh) loads ONE float4 holding k = k0 + 4*hh .. +3
// and feeds its 4 components to 4 consecutive MFMAs whose B operand uses the same k -- a 16-byte
// load per lane per 4 MFMAs and no LDS traffic for A.
//
// Software pipeline: the B chunk c+1 (float4 global loads, all issued back to back, branch-free:
// out-of-range pieces read a clamped address and are zeroed by a select) and the A fragments of
// chunk c+1 are fetched into registers while the MFMAs of chunk c run out of LDS buffer c&1; the
// registers are then written to buffer (c+1)&1 -- one barrier per chunk, no load on the MFMA path.
#pragma once
#include "common.hpp"

namespace temp {

#define GEMM_KC 40

template <int NT>
struct PanelCfg {
  static constexpr int BN = NT * 32;
  static constexpr int LDS_B = BN + 1;                       // odd row stride: conflict-free b32 reads and transposed writes
  static constexpr int NV = (GEMM_KC * BN / 4 + 255) / 256;  // float4 pieces of a B chunk per thread
};

// Piece p of B chunk [k0, k0+KC) x [n0, n0+BN) owned by this thread: in range? / global offset.
template <int NT>
__device__ __forceinline__ bool panel_b_piece(int i, int trans_b, int k0, int K, int n0, int N, int ldb, size_t* off) {
  constexpr int BN = PanelCfg<NT>::BN;
  const int p = threadIdx.x + i * 256;
  if (!trans_b) {                                            // B[k][n]: pieces run along n
    const int k = p / (BN / 4), j = (p - k * (BN / 4)) * 4;
    *off = (size_t)(k0 + k) * ldb + n0 + j;
    return (p < GEMM_KC * BN / 4) && (k0 + k < K) && (n0 + j < N);
  }
  const int j = p / (GEMM_KC / 4), k = (p - j * (GEMM_KC / 4)) * 4;   // B stored [n][k]: pieces run along k
  *off = (size_t)(n0 + j) * ldb + k0 + k;
  return (p < GEMM_KC * BN / 4) && (k0 + k < K) && (n0 + j < N);
}

// Fetch: raw float4 loads (out-of-range pieces read element 0); zero-filled by panel_store_b, i.e.
// AFTER the MFMAs of the current chunk -- a select at issue would make the wave wait for the load.
template <int NT>
__device__ __forceinline__ void panel_fetch_b(float4 (&reg)[PanelCfg<NT>::NV], const float* __restrict__ B, int ldb, int trans_b,
                                              int k0, int K, int n0, int N) {
#pragma unroll
  for (int i = 0; i < PanelCfg<NT>::NV; ++i) {
    size_t off;
    const bool ok = panel_b_piece<NT>(i, trans_b, k0, K, n0, N, ldb, &off);
    reg[i] = ld4(B + (ok ? off : 0));
  }
}

template <int NT>
__device__ __forceinline__ void panel_store_b(const float4 (&reg)[PanelCfg<NT>::NV], float* __restrict__ Bs, int ldb, int trans_b,
                                              int k0, int K, int n0, int N) {
  constexpr int BN = PanelCfg<NT>::BN, LDS_B = PanelCfg<NT>::LDS_B;
#pragma unroll
  for (int i = 0; i < PanelCfg<NT>::NV; ++i) {
    const int p = threadIdx.x + i * 256;
    size_t off;
    const bool ok = panel_b_piece<NT>(i, trans_b, k0, K, n0, N, ldb, &off);
    const float4 v = ok ? reg[i] : zero4();
    if (p < GEMM_KC * BN / 4) {
      if (!trans_b) {
        const int k = p / (BN / 4), j = (p - k * (BN / 4)) * 4;
        float* d = Bs + k * LDS_B + j;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      } else {
        const int j = p / (GEMM_KC / 4), k = (p - j * (GEMM_KC / 4)) * 4;
        float* d = Bs + k * LDS_B + j;
        d[0] = v.x; d[LDS_B] = v.y; d[2 * LDS_B] = v.z; d[3 * LDS_B] = v.w;
      }
    }
  }
}

// Up to PANEL_MAXP independent problems that share N, K, leading dimensions and the transposition
// flag run as ONE launch (blockIdx.z selects the problem): the forward and backward chains of the
// bidirectional window walk the same position at the same time, so their per-position GEMMs are
// launched together to double the waves in flight.
#define PANEL_MAXP 4
template <class Epi>
struct PanelProblem {
  int M; const float* A; const int32_t* a_idx; const float* B; Epi epi;
  const unsigned* a_keys = nullptr;        // (gemm_hx.hpp) per SOURCE row of A the key of its largest magnitude; nullptr: the launcher takes them
};
template <class Epi>
struct PanelBatch { PanelProblem<Epi> p[PANEL_MAXP]; };

// An epilogue that declares `k_split_tag` splits K over the blocks: blockIdx.y = k-slice * col_blocks + column block, the slice
// covers k in [slice * kc, slice * kc + kc) and the epilogue stores the slice's partial product (epi.kc, epi.col_blocks).
template <class E, class = void>
struct EpiKSplit { static constexpr bool value = false; };
template <class E>
struct EpiKSplit<E, decltype((void)E::k_split_tag)> { static constexpr bool value = true; };

template <int NT, class Epi>
__global__ void __launch_bounds__(256) k_gemm_panel(PanelBatch<Epi> batch, int N, int K, int lda, int ldb, int trans_b, int n_base) {
  constexpr int LDS_B = PanelCfg<NT>::LDS_B, NV = PanelCfg<NT>::NV, NQ = GEMM_KC / 8;
  __shared__ float Bs[2][GEMM_KC * LDS_B];
  const PanelProblem<Epi>& pb = batch.p[blockIdx.z];
  const int M = pb.M;
  if ((int)blockIdx.x * 128 >= M) return;                    // whole block out of range (uniform)
  const float* __restrict__ A = pb.A;
  const int32_t* __restrict__ a_idx = pb.a_idx;
  const float* __restrict__ B = pb.B;
  const Epi& epi = pb.epi;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  int col_block = blockIdx.y;
  if constexpr (EpiKSplit<Epi>::value) {                      // this block's k-slice: shift the operands, shorten K
    const int ks = blockIdx.y / epi.col_blocks, kb = ks * epi.kc;
    col_block -= ks * epi.col_blocks;
    A += kb;
    B += trans_b ? (size_t)kb : (size_t)kb * ldb;
    K = min(K - kb, epi.kc);
  }
  const int n0 = n_base + col_block * PanelCfg<NT>::BN;
  const int arow = m0 + li;
  long arow_src = -1;
  if (arow < M) arow_src = a_idx ? (long)a_idx[arow] : (long)arow;
  const bool arow_ok = arow_src >= 0;                         // invalid row: read row 0, zero by select
  const float* aptr = A + (size_t)(arow_ok ? arow_src : 0) * lda + 4 * hh;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto fetch_a = [&](float4 (&av)[NQ], int k0) {               // raw loads; select_a zero-fills at first use
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool ok = arow_ok && (k0 + q * 8 + 4 * hh < K);
      av[q] = ld4(aptr + (ok ? k0 + q * 8 : -4 * hh));           // !ok: re-read k = 0..3 of the row (in bounds)
    }
  };
  auto select_a = [&](float4 (&dst)[NQ], const float4 (&src)[NQ], int k0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) dst[q] = (arow_ok && (k0 + q * 8 + 4 * hh < K)) ? src[q] : zero4();
  };

  float4 breg[NV], av[NQ], av_next[NQ];
  panel_fetch_b<NT>(breg, B, ldb, trans_b, 0, K, n0, N);
  fetch_a(av_next, 0);
  panel_store_b<NT>(breg, Bs[0], ldb, trans_b, 0, K, n0, N);
  select_a(av, av_next, 0);
  __syncthreads();
  const int nchunks = (K + GEMM_KC - 1) / GEMM_KC;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    panel_fetch_b<NT>(breg, B, ldb, trans_b, (c + 1) * GEMM_KC, K, n0, N);   // unconditional: past the end every piece is a
    fetch_a(av_next, (c + 1) * GEMM_KC);                                     // clamped re-read, and no value merge forces a wait
    __builtin_amdgcn_sched_barrier(0);
    const float* bs = Bs[c & 1];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float as[4] = {av[q].x, av[q].y, av[q].z, av[q].w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* brow = bs + (q * 8 + 4 * hh + s) * LDS_B + li;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(brow[t * 32], as[s], acc[t], 0, 0, 0);   // operands swapped: C^T tile
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      panel_store_b<NT>(breg, Bs[(c + 1) & 1], ldb, trans_b, (c + 1) * GEMM_KC, K, n0, N);
      select_a(av, av_next, (c + 1) * GEMM_KC);
    }
    __syncthreads();
  }
  // The MFMA operands are swapped (weights as the A operand, activations as B), so the accumulator
  // tile is C^T: lane (li, hh) holds ONE output row (m0 + li) and, in registers 4q..4q+3, the four
  // consecutive columns n0 + t*32 + 8q + 4hh .. +3  ->  float4 loads/stores along the row, one row
  // mask / row scale per lane.  Two passes (all loads, then all stores).
  const int row = m0 + li;
  const bool row_ok = row < M;
  const typename Epi::RowCtx rc = epi.row_ctx(row_ok ? row : 0);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float4 pre[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = n0 + t * 32 + 8 * q + 4 * hh;
      ok[q] = row_ok && col < N;
      pre[q] = epi.pre4(rc, ok[q] ? row : 0, ok[q] ? col : 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = n0 + t * 32 + 8 * q + 4 * hh;
      if (ok[q]) epi.fin4(rc, row, col, make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]), pre[q]);
    }
  }
}

template <int NT, class Epi>
static inline void launch_panel_nt(int kid, const PanelBatch<Epi>& batch, int count, int max_m, int N, int K, int lda, int ldb, int trans_b,
                                   int n_base, int col_blocks, hipStream_t st) {
  dim3 grid(ceil_div(max_m, 128), col_blocks, count);
  TEMP_LAUNCH(kid, (k_gemm_panel<NT, Epi>), grid, dim3(256), 0, st, batch, N, K, lda, ldb, trans_b, n_base);
}

template <class Epi>
int launch_gemm_stream_multi(int kid, const PanelBatch<Epi>& batch, int count, int N, int K, int lda, int ldb, int trans_b, hipStream_t st) {
  if (count <= 0 || count > PANEL_MAXP) return TEMP_E_BADARG;
  int max_m = 0;
  long long sum_blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (batch.p[i].M > max_m) max_m = batch.p[i].M;
    sum_blocks += ceil_div(batch.p[i].M > 0 ? batch.p[i].M : 0, 128);
  }
  if (max_m <= 0 || N <= 0) return TEMP_OK;
  if (K % 4 != 0 || lda % 4 != 0 || ldb % 4 != 0 || N % 4 != 0) return TEMP_E_UNSUPPORTED;
  const int ntiles = ceil_div(N, 32);
  // column-block width: 4 tiles when the grid still fills the chip (re-reads of A come from L2),
  // narrower blocks for short panels (the per-position GEMMs of the GRU chain) to get more waves.
  int nt = 4;
  while (nt > 1 && sum_blocks * ceil_div(ntiles, nt) < 384) nt >>= 1;
  const int full = ntiles / nt, rem = ntiles - full * nt;
#define TEMP_PANEL(NT_, BASE, CB) launch_panel_nt<NT_, Epi>(kid, batch, count, max_m, N, K, lda, ldb, trans_b, (BASE), (CB), st)
  if (full > 0) {
    if (nt == 4) TEMP_PANEL(4, 0, full); else if (nt == 2) TEMP_PANEL(2, 0, full); else TEMP_PANEL(1, 0, full);
  }
  if (rem > 0) {
    const int base = full * nt * 32;
    if (rem == 3) TEMP_PANEL(3, base, 1); else if (rem == 2) TEMP_PANEL(2, base, 1); else TEMP_PANEL(1, base, 1);
  }
#undef TEMP_PANEL
  return launch_status();
}

}  // namespace temp
