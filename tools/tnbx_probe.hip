// Probe: k_gemm_tn_bx (weight-gradient product as six bf16 MFMA products) against the fp32 kernel k_gemm_tn<7,8,2>.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I temp_amd/csrc -I include tools/tnbx_probe.hip -o tools/build/tnbx_probe
#include "common.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
#include <type_traits>
#include "gemm_tn_bx.hpp"
using namespace temp;
int temp::trace_open(int, hipStream_t) { return -1; }
void temp::trace_close(int, hipStream_t) {}
bx_u32x4* temp::bx_scratch(hipStream_t, size_t) { return nullptr; }

template <class F>
float time_ms(F f, int iters = 20) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

static void run_case(int M, int Ka, int Nb, bool bias) {
  const bool w8 = tn_bx8_ok(Ka);
  const int kab = w8 ? (Ka + 255) / 256 : (Ka + 127) / 128;
  int S = w8 ? tn_bx8_slices(M, kab) : tn_bx_slices(M, kab);
  int rps = (M + S - 1) / S; rps = (rps + 15) / 16 * 16;
  float *A, *B, *part, *bpart;
  (void)hipMalloc(&A, (size_t)M * Ka * 4); (void)hipMalloc(&B, (size_t)M * Nb * 4);
  (void)hipMalloc(&part, (size_t)S * Ka * Nb * 4); (void)hipMalloc(&bpart, (size_t)S * Ka * 4);
  std::vector<float> ha((size_t)M * Ka), hb((size_t)M * Nb);
  unsigned st = 777u + M;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : ha) v = rnd() * expf(3.f * rnd());
  for (auto& v : hb) v = rnd() * 2.f;
  (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  const int nt = (Nb + 31) / 32;
  const int grid = 8 * ((S + 7) / 8) * kab;
  auto run = [&]() {
    if (w8 && nt == 7) hipLaunchKernelGGL((k_gemm_tn_bx8<7>), dim3(grid), dim3(TNBX_THREADS), 0, 0, M, Ka, Nb, A, Ka, B, Nb, rps, kab, S, part, bias ? bpart : nullptr);
    else if (w8 && nt == 5) hipLaunchKernelGGL((k_gemm_tn_bx8<5>), dim3(grid), dim3(TNBX_THREADS), 0, 0, M, Ka, Nb, A, Ka, B, Nb, rps, kab, S, part, bias ? bpart : nullptr);
    else if (nt == 7) hipLaunchKernelGGL((k_gemm_tn_bx<7>), dim3(grid), dim3(TNBX_THREADS), 0, 0, M, Ka, Nb, A, Ka, B, Nb, rps, kab, S, part, bias ? bpart : nullptr);
    else if (nt == 6) hipLaunchKernelGGL((k_gemm_tn_bx<6>), dim3(grid), dim3(TNBX_THREADS), 0, 0, M, Ka, Nb, A, Ka, B, Nb, rps, kab, S, part, bias ? bpart : nullptr);
    else hipLaunchKernelGGL((k_gemm_tn_bx<5>), dim3(grid), dim3(TNBX_THREADS), 0, 0, M, Ka, Nb, A, Ka, B, Nb, rps, kab, S, part, bias ? bpart : nullptr);
  };
  const float t = time_ms(run);
  if (w8 && nt == 7 && M == 58000) {
#define ABL8(V) { auto f = [&]() { hipLaunchKernelGGL((k_gemm_tn_bx8<7, V>), dim3(grid), dim3(TNBX_THREADS), 0, 0, M, Ka, Nb, A, Ka, B, Nb, rps, kab, S, part, bias ? bpart : nullptr); }; printf("  VAR %3d: %.4f ms\n", V, time_ms(f)); }
    ABL8(1) ABL8(4) ABL8(5) ABL8(7) ABL8(16) ABL8(20) ABL8(23) ABL8(8)
    run(); (void)hipDeviceSynchronize();
  }
  std::vector<float> hp((size_t)S * Ka * Nb), hbp((size_t)S * Ka);
  (void)hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hbp.data(), bpart, hbp.size() * 4, hipMemcpyDeviceToHost);
  // check sampled outputs against fp64
  double emax = 0, ebias = 0;
  for (int smp = 0; smp < 400; ++smp) {
    const int ka = (int)(((long long)smp * 7919 + (smp < 50 ? 0 : 13)) % Ka), nb = smp < 40 ? (smp < 20 ? smp : Nb - 1 - (smp - 20)) : (int)(((long long)smp * 104729) % Nb);
    double ref = 0, sabs = 0, got = 0;
    for (int m = 0; m < M; ++m) { const double a = ha[(size_t)m * Ka + ka], b = hb[(size_t)m * Nb + nb]; ref += a * b; sabs += fabs(a * b); }
    for (int s = 0; s < S; ++s) got += hp[(size_t)s * Ka * Nb + (size_t)ka * Nb + nb];
    const double e = fabs(got - ref) / sabs;
    if (e > emax) emax = e;
    if (bias && smp < 100) {
      double rb = 0, sb = 0, gb = 0;
      for (int m = 0; m < M; ++m) { rb += ha[(size_t)m * Ka + ka]; sb += fabs(ha[(size_t)m * Ka + ka]); }
      for (int s = 0; s < S; ++s) gb += hbp[(size_t)s * Ka + ka];
      const double eb = fabs(gb - rb) / sb;
      if (eb > ebias) ebias = eb;
    }
  }
  const double gf = 2.0 * M * Ka * Nb / 1e9;
  printf("M=%6d Ka=%3d Nb=%3d S=%d kab=%d  bx %.4f ms %6.1f TF | err/sum|ab| %.2e  bias err %.2e\n", M, Ka, Nb, S, kab, t, gf / t, emax, ebias);
  (void)hipFree(A); (void)hipFree(B); (void)hipFree(part); (void)hipFree(bpart);
}

int main() {
  run_case(58000, 600, 200, true);
  run_case(82000, 200, 200, true);
  run_case(58003, 600, 200, false);
  run_case(20000, 600, 136, true);
  run_case(30000, 328, 200, true);
  return 0;
}
