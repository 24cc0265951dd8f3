// Row-panel fp32 MFMA GEMM template shared by gemm_kernels.hip and gru_kernels.hip.
// A wave owns 32 rows x (NT*32) columns; the 4 waves of a block own 4 consecutive row tiles and
// share the B chunk staged in LDS.  A is read straight from global memory: the MFMA sums over k in
// any order, so lane (row i, half hh) loads ONE float4 holding k = k0 + 4*hh .. +3 and feeds its 4
// components to 4 consecutive MFMAs whose B operand uses the same k -- a 16-byte load per lane per
// 4 MFMAs and no LDS traffic for A.
#pragma once
#include "common.hpp"

namespace temp {

#define GEMM_KC 40

template <int NT, class Epi>
__global__ void __launch_bounds__(256) k_gemm_panel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                    const int32_t* __restrict__ a_idx, const float* __restrict__ B, int ldb,
                                                    int trans_b, Epi epi) {
  constexpr int BN = NT * 32, LDS_B = BN + 1;
  __shared__ float Bs[GEMM_KC * LDS_B];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  const int n0 = blockIdx.y * BN;
  const int arow = m0 + li;
  // Branch-free inner loop: an invalid row reads row 0 (always mapped when M > 0) and is zeroed by a select.
  long arow_src = -1;
  if (arow < M) arow_src = a_idx ? (long)a_idx[arow] : (long)arow;
  const bool arow_ok = arow_src >= 0;
  const float* aptr = A + (size_t)(arow_ok ? arow_src : 0) * lda + 4 * hh;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int k0 = 0; k0 < K; k0 += GEMM_KC) {
    const int kc = min(GEMM_KC, K - k0);
    __syncthreads();
    // stage B[k0 .. k0+KC) x [n0 .. n0+BN); rows >= kc and columns >= N are zero so the MFMA loop needs no guards
    if (!trans_b) {
      for (int idx = threadIdx.x; idx < GEMM_KC * BN; idx += 256) {
        const int k = idx / BN, j = idx - k * BN;
        Bs[k * LDS_B + j] = (k < kc && n0 + j < N) ? B[(size_t)(k0 + k) * ldb + n0 + j] : 0.f;
      }
    } else {
      for (int idx = threadIdx.x; idx < GEMM_KC * BN; idx += 256) {
        const int j = idx / GEMM_KC, k = idx - j * GEMM_KC;
        Bs[k * LDS_B + j] = (k < kc && n0 + j < N) ? B[(size_t)(n0 + j) * ldb + k0 + k] : 0.f;
      }
    }
    __syncthreads();
    float4 av[GEMM_KC / 8];
#pragma unroll
    for (int q = 0; q < GEMM_KC / 8; ++q) {          // all A loads of the chunk issued up front
      const bool ok = arow_ok && (q * 8 + 4 * hh < kc);
      const float4 v = ld4(aptr + (ok ? k0 + q * 8 : -4 * hh));     // !ok: re-read k=0..3 of the row (in bounds)
      av[q] = ok ? v : zero4();
    }
#pragma unroll
    for (int q = 0; q < GEMM_KC / 8; ++q) {
      const float as[4] = {av[q].x, av[q].y, av[q].z, av[q].w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* brow = Bs + (q * 8 + 4 * hh + s) * LDS_B + li;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], brow[t * 32], acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + t * 32 + li;
    if (col < N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row < M) epi(row, col, acc[t][r]);
      }
    }
  }
}

template <class Epi>
int launch_gemm_panel(int kid, int M, int N, int K, const float* A, int lda, const int32_t* a_idx, const float* B, int ldb, int trans_b,
                      const Epi& epi, hipStream_t st) {
  if (M <= 0 || N <= 0) return TEMP_OK;
  if (K % 4 != 0 || lda % 4 != 0) return TEMP_E_UNSUPPORTED;
  const int row_blocks = ceil_div(M, 128);
  const int ntiles = ceil_div(N, 32);
  // widest column block that still leaves >= ~2 blocks per CU; narrow blocks re-read A from L2.
  int nt = 7;
  while (nt > 1 && (long long)row_blocks * ceil_div(ntiles, nt) < 512) nt = (nt == 7) ? 4 : nt / 2;
  dim3 grid(row_blocks, ceil_div(ntiles, nt));
  switch (nt) {
    case 7: TEMP_LAUNCH(kid, (k_gemm_panel<7, Epi>), grid, dim3(256), 0, st, M, N, K, A, lda, a_idx, B, ldb, trans_b, epi); break;
    case 4: TEMP_LAUNCH(kid, (k_gemm_panel<4, Epi>), grid, dim3(256), 0, st, M, N, K, A, lda, a_idx, B, ldb, trans_b, epi); break;
    case 2: TEMP_LAUNCH(kid, (k_gemm_panel<2, Epi>), grid, dim3(256), 0, st, M, N, K, A, lda, a_idx, B, ldb, trans_b, epi); break;
    default: TEMP_LAUNCH(kid, (k_gemm_panel<1, Epi>), grid, dim3(256), 0, st, M, N, K, A, lda, a_idx, B, ldb, trans_b, epi); break;
  }
  return launch_status();
}


}  // namespace temp
