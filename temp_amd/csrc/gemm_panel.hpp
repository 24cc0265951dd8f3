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
  const bool arow_ok = arow < M;
  const float* aptr = nullptr;
  if (arow_ok) {
    const long r = a_idx ? (long)a_idx[arow] : (long)arow;
    aptr = (r >= 0) ? A + (size_t)r * lda : nullptr;
  }
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  int ntv = (N - n0 + 31) / 32;
  if (ntv > NT) ntv = NT;

  for (int k0 = 0; k0 < K; k0 += GEMM_KC) {
    const int kc = min(GEMM_KC, K - k0);
    __syncthreads();
    if (!trans_b) {
      for (int idx = threadIdx.x; idx < kc * BN; idx += 256) {
        const int k = idx / BN, j = idx - k * BN;
        Bs[k * LDS_B + j] = (n0 + j < N) ? B[(size_t)(k0 + k) * ldb + n0 + j] : 0.f;
      }
    } else {
      for (int idx = threadIdx.x; idx < kc * BN; idx += 256) {
        const int j = idx / kc, k = idx - j * kc;
        Bs[k * LDS_B + j] = (n0 + j < N) ? B[(size_t)(n0 + j) * ldb + k0 + k] : 0.f;
      }
    }
    __syncthreads();
    for (int kk = 0; kk < kc; kk += 8) {
      const int kb = kk + 4 * hh;          // this lane's 4 k values inside the chunk
      float4 a = zero4();
      if (aptr && kb < kc) a = ld4(aptr + k0 + kb);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int krow = kb + s;
        const bool kok = krow < kc;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t < ntv) {
            const float b = kok ? Bs[krow * LDS_B + t * 32 + li] : 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < ntv) {
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
