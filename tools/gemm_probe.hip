// Ablation probe for the row-panel MFMA GEMM (development tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itemp_amd/csrc tools/gemm_probe.hip -o gpurun_out/gemm_probe
#include "common.hpp"
#include "gemm_wres.hpp"
#include <cstdio>
#include <vector>
using namespace temp;

int temp::trace_open(int, hipStream_t) { return -1; }
void temp::trace_close(int, hipStream_t) {}

struct EpiStoreP {
  float* out; int ldo;
  __device__ __forceinline__ float pre(int, int) const { return 0.f; }
  __device__ __forceinline__ void fin(int row, int col, float acc, float) const { out[(size_t)row * ldo + col] = acc; }
};

// pure MFMA issue-rate check: 4 independent accumulators per wave
__global__ void __launch_bounds__(256) k_mfma_peak(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// VAR bit0: skip B re-staging (reuse chunk 0), bit1: skip A loads (constant), bit2: skip epilogue stores (write 1 value)
template <int NT, int VAR>
__global__ void __launch_bounds__(256) k_panel_var(int M, int N, int K, const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                   int ldb, int trans_b, int n_base, float* out, int ldo) {
  constexpr int LDS_B = PanelCfg<NT>::LDS_B, NV = PanelCfg<NT>::NV, NQ = GEMM_KC / 8;
  __shared__ float Bs[2][GEMM_KC * LDS_B];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  const int n0 = n_base + blockIdx.y * PanelCfg<NT>::BN;
  const int arow = m0 + li;
  const bool arow_ok = arow < M;
  const float* aptr = A + (size_t)(arow_ok ? arow : 0) * lda + 4 * hh;
  f32x16 acc[NT];
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto fetch_a = [&](float4 (&av)[NQ], int k0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (VAR & 2) { av[q] = make_float4(1.f, 2.f, 3.f, 4.f); continue; }
      const bool ok = arow_ok && (k0 + q * 8 + 4 * hh < K);
      const float4 v = ld4(aptr + (ok ? k0 + q * 8 : -4 * hh));
      av[q] = ok ? v : zero4();
    }
  };
  float4 breg[NV], av[NQ], av_next[NQ];
  panel_fetch_b<NT>(breg, B, ldb, trans_b, 0, K, n0, N);
  fetch_a(av, 0);
  panel_store_b<NT>(breg, Bs[0], trans_b);
  if (VAR & 1) panel_store_b<NT>(breg, Bs[1], trans_b);
  __syncthreads();
  const int nchunks = (K + GEMM_KC - 1) / GEMM_KC;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) {
      if (!(VAR & 1)) panel_fetch_b<NT>(breg, B, ldb, trans_b, (c + 1) * GEMM_KC, K, n0, N);
      fetch_a(av_next, (c + 1) * GEMM_KC);
    }
    const float* bs = Bs[c & 1];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float as[4] = {av[q].x, av[q].y, av[q].z, av[q].w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* brow = bs + (q * 8 + 4 * hh + s) * LDS_B + li;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], brow[t * 32], acc[t], 0, 0, 0);
      }
    }
    if (more) {
      if (!(VAR & 1)) panel_store_b<NT>(breg, Bs[(c + 1) & 1], trans_b);
#pragma unroll
      for (int q = 0; q < NQ; ++q) av[q] = av_next[q];
    }
    if (!(VAR & 1)) __syncthreads();
  }
  if (VAR & 4) {
    float s = 0.f;
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (m0 + li < M) out[(size_t)(m0 + li) * ldo + n0 + hh] = s;
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + t * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (col < N && row < M) out[(size_t)row * ldo + col] = acc[t][r];
    }
  }
}

template <class F>
float time_ms(F f, int iters = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const int M = 116000, K = 200, N = 200;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)M * 600 * 4); hipMalloc(&B, (size_t)600 * 600 * 4); hipMalloc(&C, (size_t)M * 600 * 4);
  std::vector<float> h((size_t)M * 600);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)600 * 600 * 4, hipMemcpyHostToDevice);
  {
    const int iters = 2000;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_peak, dim3(2048), dim3(256), 0, 0, C, iters); }, 5);
    double fl = 2048.0 * 4 * iters * 4 * 4096.0;
    printf("mfma_peak: %.3f ms  %.1f TF/s\n", ms, fl / ms / 1e9);
  }
  const double gf = 2.0 * M * K * 128 / 1e9;
#define RUN(NT_, VAR_, name)                                                                                              \
  {                                                                                                                         \
    float ms = time_ms([&] { hipLaunchKernelGGL((k_panel_var<NT_, VAR_>), dim3((M + 127) / 128, 1), dim3(256), 0, 0, M, N, K, A, 200, B, 200, 0, 0, C, 200); }); \
    printf("%-34s NT=%d  %.4f ms  %.1f TF/s (useful, %d cols)\n", name, NT_, ms, 2.0 * M * K * (NT_ * 32) / ms / 1e9, NT_ * 32);                               \
  }
  RUN(4, 0, "full")
  RUN(4, 1, "no B restage/no barrier")
  RUN(4, 2, "no A loads")
  RUN(4, 4, "no epilogue")
  RUN(4, 7, "mfma+lds only")
  RUN(4, 3, "no A, no B restage")
  RUN(2, 0, "full")
  RUN(2, 7, "mfma+lds only")
  RUN(1, 0, "full")
  (void)gf;
  return 0;
}
