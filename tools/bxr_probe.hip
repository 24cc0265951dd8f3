// Probe: k_gemm_bxr (weights-resident split-operand GEMM, gemm_bxr.hpp) -- ablations and per-wave s_memtime stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBXR_PROBE -I temp_amd/csrc -I include tools/bxr_probe.hip -o tools/build/bxr_probe
#include "common.hpp"
#include "gemm_wres.hpp"
#include "gemm_bx.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
using namespace temp;
int temp::trace_open(int, hipStream_t) { return -1; }
static bx_u32x4* g_scr = nullptr;
bx_u32x4* temp::bx_scratch(hipStream_t, size_t bytes) { if (!g_scr) (void)hipMalloc(&g_scr, BX_SLOT_BYTES); return bytes <= BX_SLOT_BYTES ? g_scr : nullptr; }
void temp::trace_close(int, hipStream_t) {}

struct EpiAddP {                       // out = relu(acc + addend + bias): the self-loop epilogue (the library's EpiAddBiasAct)
  const float* addend; const float* bias; float* out; int ldo;
  struct RowCtx { int add; };
  __device__ __forceinline__ RowCtx row_ctx(int) const { RowCtx c; c.add = addend ? 1 : 0; return c; }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int row, int col) const { return add4(ld4(addend + (size_t)row * ldo + col), ld4(bias + col)); }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4 p) const {
    st4(out + (size_t)row * ldo + col, make_float4(fmaxf(acc.x + p.x, 0.f), fmaxf(acc.y + p.y, 0.f), fmaxf(acc.z + p.z, 0.f), fmaxf(acc.w + p.w, 0.f)));
  }
  static constexpr int k_raw_pre = 1;
  __device__ __forceinline__ bool has_addend() const { return addend != nullptr; }
  __device__ __forceinline__ bool has_row_mask() const { return false; }
  __device__ __forceinline__ float4 raw4(int row, int col) const { return ld4(addend + (size_t)row * ldo + col); }
  __device__ __forceinline__ float bias1(int col) const { return bias ? bias[col] : 0.f; }
};
#ifndef NO_INIT
namespace temp { template <> struct EpiAccInit<EpiAddP> { static constexpr bool value = true; }; }
#endif

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

template <int VAR>
static float run_var(const PanelBatch<EpiAddP>& b, const BxrGeom& rg, const BxPacked& pk, bool stamps) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_bxr<EpiAddP, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, BXR_LDS_BYTES);
  const size_t lds = (size_t)rg.n_slabs * BXR_G * 192 * 16 + BXR_BIAS_BYTES;
  auto f = [&]() { hipLaunchKernelGGL((k_gemm_bxr<EpiAddP, VAR>), dim3(256, 1), dim3(BXR_WAVES * 64), lds, 0, b, rg, pk); };
  const float t = time_ms(f);
  printf("  VAR %2d: %.4f ms\n", VAR, t);
  if (stamps) {
    std::vector<unsigned long long> h(256 * 8 * 4), he(256 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_bxr_stamp), h.size() * 8);
    (void)hipMemcpyFromSymbol(he.data(), HIP_SYMBOL(g_bxr_epi), he.size() * 8);
    { std::vector<unsigned long long> z(256 * 8, 0); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bxr_epi), z.data(), z.size() * 8); }
    {
      std::vector<unsigned long long> rt(256 * 8 * 2);
      (void)hipMemcpyFromSymbol(rt.data(), HIP_SYMBOL(g_bxr_rt), rt.size() * 8);
      double cyc = 0, ns = 0;
      for (int i = 0; i < 2048; ++i) { if (h[4 * i + 3] == 0) continue; cyc += (double)(h[4 * i + 2] - h[4 * i]); ns += 10.0 * (double)(rt[2 * i + 1] - rt[2 * i]); }
      printf("    s_memtime ticks per ns of s_memrealtime (100 MHz) over the waves' lifetimes: %.3f (= GHz if s_memtime is the shader clock)\n", cyc / ns);
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < 2048; ++i) { if (h[4 * i + 3] == 0) continue; t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 2]); }
    printf("    kernel span %llu ticks\n", t1 - t0);
    for (int gt = 1; gt <= 4; ++gt)
      for (int np = 1; np <= 6; ++np) {
        double st = 0, run = 0, start = 0, mx = 0, ep = 0; int n = 0;
        for (int i = 0; i < 2048; ++i) {
          if ((int)(h[4 * i + 3] >> 32) != gt || (int)(h[4 * i + 3] & 0xffffffffu) != np) continue;
          ++n; ep += (double)he[i] / 21.0; st += (double)(h[4 * i + 1] - h[4 * i]); run += (double)(h[4 * i + 2] - h[4 * i + 1]); start += (double)(h[4 * i] - t0);
          mx = std::max(mx, (double)(h[4 * i + 2] - t0));
        }
        if (n) printf("    tiles %d panels %d: %4d waves | start +%7.0f | staging %7.0f | panels %8.0f (%7.0f per panel) | epilogue (stores issued, excl. last) %7.0f per panel\n", gt, np, n, start / n, st / n, run / n, run / n / np, ep / n / (np > 1 ? np - 1 : 1));
      }
  }
  return t;
}

static void run_case(int M, int K, int N) {
  float *A, *B, *C, *ADD, *BIAS;
  (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&B, (size_t)N * K * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
  (void)hipMalloc(&ADD, (size_t)M * N * 4); (void)hipMalloc(&BIAS, N * 4);
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  unsigned st = 12345u + M + K;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : ha) v = rnd();
  for (auto& v : hb) v = rnd() * 0.3f;
  (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(ADD, 0, (size_t)M * N * 4); (void)hipMemset(BIAS, 0, N * 4);
  PanelBatch<EpiAddP> b;
  for (int i = 0; i < PANEL_MAXP; ++i) b.p[i] = PanelProblem<EpiAddP>{0, nullptr, nullptr, nullptr, EpiAddP{ADD, BIAS, C, N}};
  b.p[0] = PanelProblem<EpiAddP>{M, A, nullptr, B, EpiAddP{ADD, BIAS, C, N}};
  BxGeom bg; int G;
  if (!bx_plan(N, K, K, K, 1, M, M, &bg, &G)) { printf("bx_plan refused\n"); return; }
  BxPacked pk;
  if (!bx_pack_batch(b, 1, bg, 0, &pk)) { printf("pack refused\n"); return; }
  BxrGeom rg;
  if (!bxr_plan(N, K, K, M, &rg)) { printf("bxr_plan refused\n"); return; }
  printf("M=%d K=%d N=%d: %d tiles, %d groups, slots", M, K, N, rg.n_tiles, rg.n_groups);
  for (int j = 0; j < rg.n_groups; ++j) printf(" %d", rg.group_slots[j]);
  const double mfma = (double)ceil_div(M, 32) * rg.n_tiles * rg.n_slabs * 6 * 32 / 1024.0;
  printf(", %d panels per XCD; MFMA issue bound %.0f cycles per SIMD\n", rg.per_xcd, mfma);
  run_var<0>(b, rg, pk, true);
  if (rg.n_tiles == 19) {                                     // alternative split for the input-gate shape: (4,3,3,3,3,3) tiles on (7,5,5,5,5,5) slots
    BxrGeom r2 = rg;
    const int tl[6] = {4, 3, 3, 3, 3, 3}, sl[6] = {7, 5, 5, 5, 5, 5};
    r2.n_groups = 6;
    int sidx = 0, t = 0;
    for (int j = 0; j < 6; ++j) {
      r2.group_t0[j] = (unsigned char)t; r2.group_nt[j] = (unsigned char)tl[j]; t += tl[j];
      r2.group_slots[j] = (unsigned char)sl[j];
      for (int r = 0; r < sl[j]; ++r, ++sidx) { r2.slot_group[sidx] = (unsigned char)j; r2.slot_rank[sidx] = (unsigned char)r; }
    }
    printf("  alternative split (4,3,3,3,3,3) x (7,5,5,5,5,5) [measured: 136.8 against 137.9 us, no gain]:\n");
    run_var<0>(b, r2, pk, false);
  }
  run_var<64>(b, rg, pk, true);
  run_var<128>(b, rg, pk, true);
  run_var<192>(b, rg, pk, true);
  run_var<193>(b, rg, pk, true);
  (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(ADD); (void)hipFree(BIAS);
}

int main() {
  run_case(81984, 200, 200);
  run_case(57984, 200, 600);
  return 0;
}
