// Probe: sustained issue rate of v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32 with every CU busy (clock under load).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int BF, int NACC>
__global__ void __launch_bounds__(256) k_rate(int iters, float* out, unsigned long long* clk) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float fa = threadIdx.x * 0.001f, fb = 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      if constexpr (BF) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int t = 0; t < NACC; ++t) s += acc[t][0];
  if (s == 12345.f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int BF, int NACC>
void run(const char* name, int blocks, int iters) {
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, 4); (void)hipMalloc(&clk, 16);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k_rate<BF, NACC>), dim3(blocks), dim3(256), 0, 0, iters, out, clk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k_rate<BF, NACC>), dim3(blocks), dim3(256), 0, 0, iters, out, clk);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * NACC;                  // per wave
  const double waves_per_simd = blocks / 256.0;                // 4 waves per block, 4 SIMDs per CU, 256 CUs
  const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
  printf("%-28s blocks %4d: %.3f ms  shader clock %.2f GHz  cycles/MFMA/SIMD (wall) %.1f  (wave 0: %.1f)\n", name, blocks, ms, ghz,
         ms * 1e-3 * ghz * 1e9 / (n_mfma * waves_per_simd), (double)h[0] / (n_mfma * (waves_per_simd < 1 ? 1 : waves_per_simd)));
}
int main() {
  run<1, 4>("bf16 32x32x16, 4 acc", 256, 20000);
  run<1, 4>("bf16 32x32x16, 4 acc", 512, 20000);
  run<1, 1>("bf16 32x32x16, 1 acc (dep)", 256, 40000);
  run<1, 1>("bf16 32x32x16, 1 acc (dep)", 512, 40000);
  run<0, 4>("f32 32x32x2, 4 acc", 256, 10000);
  run<0, 4>("f32 32x32x2, 4 acc", 512, 10000);
  run<0, 1>("f32 32x32x2, 1 acc (dep)", 256, 20000);
  return 0;
}
