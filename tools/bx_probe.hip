// Probe: k_gemm_bx (fp32 GEMM as six bf16 MFMA products) against the fp32 MFMA kernels: time and error vs fp64.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I temp_amd/csrc -I include tools/bx_probe.hip -o tools/build/bx_probe
#include "common.hpp"
#include "gemm_wres.hpp"
#include "gemm_bx.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace temp;
int temp::trace_open(int, hipStream_t) { return -1; }
static bx_u32x4* g_scr = nullptr;
bx_u32x4* temp::bx_scratch(hipStream_t, size_t bytes) { if (!g_scr) (void)hipMalloc(&g_scr, BX_SLOT_BYTES); return bytes <= BX_SLOT_BYTES ? g_scr : nullptr; }
void temp::trace_close(int, hipStream_t) {}

struct EpiStoreP {
  float* out; int ldo;
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row_ctx(int) const { return RowCtx(); }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int, int) const { return zero4(); }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4) const { st4(out + (size_t)row * ldo + col, acc); }
};

struct EpiAddP {                       // out = relu(acc + addend + bias): the self-loop epilogue's memory behaviour
  const float* add; const float* bias; float* out; int ldo;
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row_ctx(int) const { return RowCtx(); }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int row, int col) const { return ld4(add + (size_t)row * ldo + col); }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4 pre) const {
    const float4 b = ld4(bias + col);
    st4(out + (size_t)row * ldo + col, make_float4(fmaxf(acc.x + pre.x + b.x, 0.f), fmaxf(acc.y + pre.y + b.y, 0.f), fmaxf(acc.z + pre.z + b.z, 0.f), fmaxf(acc.w + pre.w + b.w, 0.f)));
  }
};
__global__ void k_flush(float4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f); }
static float4* g_flush = nullptr;
template <class F>
float time_cold_ms(F f, int iters = 6) {     // every run behind a 768-MB streaming write: inputs come from HBM, not from the Infinity Cache
  const size_t n = (768u << 20) / 16;
  if (!g_flush) (void)hipMalloc(&g_flush, n * 16);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float tot = 0;
  for (int i = 0; i < iters; ++i) {
    hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, g_flush, n);
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); tot += ms;
  }
  return tot / iters;
}

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

static void run_case(int M, int K, int N, int trans_b, int force_g) {
  float *A, *B, *C1, *C2;
  (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&B, (size_t)N * K * 4); (void)hipMalloc(&C1, (size_t)M * N * 4); (void)hipMalloc(&C2, (size_t)M * N * 4);
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  unsigned st = 12345u + M + K;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : ha) v = rnd() * 2.f * expf(4.f * rnd());
  for (auto& v : hb) v = rnd() * 0.3f;
  (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(C1, 0, (size_t)M * N * 4); (void)hipMemset(C2, 0, (size_t)M * N * 4);
  const int ldb = trans_b ? K : N;
  PanelBatch<EpiStoreP> b1, b2;
  for (int i = 0; i < PANEL_MAXP; ++i) { b1.p[i] = PanelProblem<EpiStoreP>{0, nullptr, nullptr, nullptr, EpiStoreP{C1, N}}; b2.p[i] = b1.p[i]; }
  b1.p[0] = PanelProblem<EpiStoreP>{M, A, nullptr, B, EpiStoreP{C1, N}};
  b2.p[0] = PanelProblem<EpiStoreP>{M, A, nullptr, B, EpiStoreP{C2, N}};
  WresGeom wg;
  const bool wres = wres_plan(N, K, K, ldb, trans_b, M, &wg);
  auto run_f32 = [&]() {
    if (wres) launch_gemm_wres(0, b1, 1, wg, 0); else launch_gemm_stream_multi(0, b1, 1, N, K, K, ldb, trans_b, 0);
  };
  BxGeom bg; int G;
  if (!bx_plan(N, K, K, ldb, trans_b, M, M, &bg, &G)) { printf("bx_plan refused\n"); return; }
  if (force_g > 0) { G = force_g; bg.n_groups = ceil_div(bg.n_tiles, G); bg.tail_store = bg.n_tiles - (bg.n_groups - 1) * G; }
  auto run_bx = [&]() { launch_gemm_bx(0, b2, 1, bg, G, 0); };
  {
    dim3 grid(8 * bg.per_xcd * bg.n_groups, 1);
    BxPacked pk; for (int i = 0; i < PANEL_MAXP; ++i) pk.b[i] = temp::bx_scratch(0, 1);
#define ABLG(GG, V) { auto f = [&]() { hipLaunchKernelGGL((k_gemm_bxp<GG, EpiStoreP, V>), grid, dim3(BX_THREADS), 0, 0, b2, bg, pk); }; printf("  VAR %2d: %.4f ms\n", V, time_ms(f)); }
#define ABL(V) { if (G == 7) ABLG(7, V) else if (G == 5) ABLG(5, V) else if (G == 4) ABLG(4, V) }
    run_bx();

  }
  const float t1 = time_ms(run_f32), t2 = time_ms(run_bx);
  {
    float *ADD, *BIAS; (void)hipMalloc(&ADD, (size_t)M * N * 4); (void)hipMalloc(&BIAS, N * 4); (void)hipMemset(ADD, 0, (size_t)M * N * 4); (void)hipMemset(BIAS, 0, N * 4);
    PanelBatch<EpiAddP> b3;
    for (int i = 0; i < PANEL_MAXP; ++i) b3.p[i] = PanelProblem<EpiAddP>{0, nullptr, nullptr, nullptr, EpiAddP{ADD, BIAS, C2, N}};
    b3.p[0] = PanelProblem<EpiAddP>{M, A, nullptr, B, EpiAddP{ADD, BIAS, C2, N}};
    auto run_add = [&]() { launch_gemm_bx(0, b3, 1, bg, G, 0); };
    printf("  bx warm %.4f | bx cold %.4f | bx + addend epilogue warm %.4f cold %.4f | fp32 cold %.4f ms\n", t2, time_cold_ms(run_bx), time_ms(run_add), time_cold_ms(run_add), time_cold_ms(run_f32));
    (void)hipFree(ADD); (void)hipFree(BIAS);
  }
  const double gf = 2.0 * M * K * N / 1e9;
  std::vector<float> c1((size_t)M * N), c2((size_t)M * N);
  (void)hipMemcpy(c1.data(), C1, c1.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost);
  double e1m = 0, e2m = 0, dm = 0;
  for (int r = 0; r < 1500; ++r) {
    const int row = r < 300 ? (r < 150 ? r : M - 1 - (r - 150)) : (int)(((long long)r * 7919) % M);
    for (int n = 0; n < N; ++n) {
      double ref = 0, sabs = 0;
      for (int k = 0; k < K; ++k) {
        const double a = ha[(size_t)row * K + k], b = trans_b ? hb[(size_t)n * K + k] : hb[(size_t)k * N + n];
        ref += a * b; sabs += fabs(a * b);
      }
      const double e1 = fabs(c1[(size_t)row * N + n] - ref) / sabs, e2 = fabs(c2[(size_t)row * N + n] - ref) / sabs;
      if (e1 > e1m) e1m = e1;
      if (e2 > e2m) e2m = e2;
      const double d = fabs((double)c1[(size_t)row * N + n] - c2[(size_t)row * N + n]) / sabs;
      if (d > dm) dm = d;
    }
  }
  printf("M=%6d K=%3d N=%3d tb=%d  fp32 %s %.4f ms %6.1f TF | bx G=%d x%d %.4f ms %6.1f TF (%.2fx) | err/sum|ab| fp32 %.2e bx %.2e diff %.2e\n", M, K, N,
         trans_b, wres ? "wres" : "strm", t1, gf / t1, G, bg.n_groups, t2, gf / t2, t1 / t2, e1m, e2m, dm);
  (void)hipFree(A); (void)hipFree(B); (void)hipFree(C1); (void)hipFree(C2);
}

int main(int argc, char** argv) {
  const int fg = argc > 1 ? atoi(argv[1]) : 0;
  run_case(58000, 200, 600, 1, fg);
  run_case(58000, 600, 200, 0, fg);
  run_case(82000, 200, 200, 0, fg);
  run_case(82000, 200, 200, 1, fg);
  return 0;
}
