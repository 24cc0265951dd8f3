// Ablation probe for the weights-resident MFMA GEMM (development tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itemp_amd/csrc tools/wres_probe.hip -o tools/wres_probe
#include "common.hpp"
#include "gemm_wres.hpp"
#include "../temp_amd/csrc/gemm_kernels.hip"
#include "../temp_amd/csrc/gru_kernels.hip"
#include <cstdio>
#include <vector>
using namespace temp;
static unsigned long long* DBG = nullptr;


struct EpiP {
  float* out; int ldo;
  unsigned long long* dbg;      // [waves][8]: first stamp of each kind, plus the last stamp 3 in slot 5
  __device__ __forceinline__ void stamp(int k, unsigned long long t) const {
    if ((threadIdx.x & 63) != 0 || !dbg) return;
    unsigned long long* d = dbg + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    if (d[k] == 0) d[k] = t;
  }
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row_ctx(int) const { return RowCtx(); }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int, int) const { return zero4(); }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4) const { st4(out + (size_t)row * ldo + col, acc); }
};

template <class F>
static float time_ms(F f, int reps = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

template <int NTS, int VAR>
static void run(const char* name, int M, int N, int K, const float* A, const float* B, float* C, int trans_b, int blocks_per_cu = 2) {
  WresGeom g;
  if (!wres_plan(N, K, K, trans_b ? K : N, trans_b, M, &g)) { printf("plan failed\n"); return; }
  g.tps = NTS;                                      // override the planner's width
  g.n_slices = (g.n_tiles + NTS - 1) / NTS;
  g.tail_store = g.n_tiles - (g.n_slices - 1) * NTS;
  PanelBatch<EpiP> batch;
  for (int i = 0; i < PANEL_MAXP; ++i) batch.p[i] = PanelProblem<EpiP>{0, nullptr, nullptr, nullptr, EpiP{C, N, DBG}};
  batch.p[0] = PanelProblem<EpiP>{M, A, nullptr, B, EpiP{C, N, DBG}};
  hipFuncSetAttribute((const void*)k_gemm_wres<NTS, EpiP, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, WRES_LDS_BYTES);
  int bps = 32 * blocks_per_cu / g.n_slices;
  if (bps < 1) bps = 1;
  const size_t lds = (size_t)NTS * 32 * g.ldk * 4;
  float ms = time_ms([&] { hipLaunchKernelGGL((k_gemm_wres<NTS, EpiP, VAR>), dim3(g.n_slices * bps * 8), dim3(256), lds, 0, batch, 1, g, bps); });
  const double fl = 2.0 * M * K * (double)N;
  printf("%-30s M=%6d N=%3d K=%3d NTS=%d slices=%d bps=%2d  %.4f ms  %.1f TF/s useful\n", name, M, N, K, NTS, g.n_slices, bps, ms, fl / ms / 1e9);
}

template <int NTS>
static void timeline(const char* name, int M, int N, int K, const float* A, const float* B, float* C, int trans_b, int blocks_per_cu) {
  WresGeom g;
  wres_plan(N, K, K, trans_b ? K : N, trans_b, 1 << 20, &g);
  g.tps = NTS; g.n_slices = (g.n_tiles + NTS - 1) / NTS; g.tail_store = g.n_tiles - (g.n_slices - 1) * NTS;
  int bps = 32 * blocks_per_cu / g.n_slices;
  const int nblk = g.n_slices * bps * 8, nw = nblk * 4;
  unsigned long long* d;
  hipMalloc(&d, (size_t)nw * 64);
  hipMemset(d, 0, (size_t)nw * 64);
  DBG = d;
  PanelBatch<EpiP> batch;
  for (int i = 0; i < PANEL_MAXP; ++i) batch.p[i] = PanelProblem<EpiP>{0, nullptr, nullptr, nullptr, EpiP{C, N, d}};
  batch.p[0] = PanelProblem<EpiP>{M, A, nullptr, B, EpiP{C, N, d}};
  hipFuncSetAttribute((const void*)k_gemm_wres<NTS, EpiP, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, WRES_LDS_BYTES);
  const size_t lds = (size_t)NTS * 32 * g.ldk * 4;
  hipLaunchKernelGGL((k_gemm_wres<NTS, EpiP, 0>), dim3(nblk), dim3(256), lds, 0, batch, 1, g, bps);      // warm
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k_gemm_wres<NTS, EpiP, 8>), dim3(nblk), dim3(256), lds, 0, batch, 1, g, bps);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)nw * 8);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  // per-wave durations (each XCD has its own counter base, so only differences inside a wave are meaningful)
  const char* lab[7] = {"prologue (start -> W slice in LDS)", "panel 1: A latency + MFMAs", "panel 1: epilogue", "rest (further panels)", "whole wave", "  prologue: start -> loads issued", "  prologue: loads issued -> LDS stores issued"};
  double sum[7] = {0}, mx[7] = {0};
  int cnt[7] = {0};
  auto add = [&](int k, unsigned long long a, unsigned long long b) { if (a && b && b >= a) { const double v = (double)(b - a); sum[k] += v; if (v > mx[k]) mx[k] = v; cnt[k]++; } };
  for (int w = 0; w < nw; ++w) {
    const unsigned long long* t = &h[(size_t)w * 8];
    add(0, t[0], t[1]); add(1, t[1], t[2]); add(2, t[2], t[3]); add(3, t[3], t[4]); add(4, t[0], t[4]); add(5, t[0], t[7]); add(6, t[7], t[5]);
  }
  printf("%s M=%d N=%d K=%d NTS=%d blocks=%d: s_memtime ticks per wave (avg / max)\n", name, M, N, K, NTS, nblk);
  for (int k = 0; k < 7; ++k) printf("   %-40s avg %9.0f  max %9.0f  (n=%d)\n", lab[k], cnt[k] ? sum[k] / cnt[k] : 0.0, mx[k], cnt[k]);
  DBG = nullptr;
  hipFree(d);
}

static void gru_timeline(int n, int D) {
  // one window-chain level: 2 cells of n rows, hoisted gi, previous state gathered through an identity map
  typedef EpiGruCell<TEMP_GRU_TORCH> Epi;
  const int cells = 2;
  float *gi, *prev, *W, *bh, *dt, *H, *saved;
  int32_t* idx;
  hipMalloc(&gi, (size_t)cells * n * 3 * D * 4); hipMalloc(&prev, (size_t)cells * n * D * 4); hipMalloc(&W, (size_t)cells * 3 * D * D * 4);
  hipMalloc(&bh, 3 * D * 4); hipMalloc(&dt, (size_t)n * 4); hipMalloc(&H, (size_t)cells * n * D * 4); hipMalloc(&saved, (size_t)5 * cells * n * D * 4);
  hipMalloc(&idx, (size_t)n * 4);
  hipMemset(gi, 0, (size_t)cells * n * 3 * D * 4); hipMemset(prev, 0, (size_t)cells * n * D * 4); hipMemset(W, 0, (size_t)cells * 3 * D * D * 4);
  hipMemset(bh, 0, 3 * D * 4); hipMemset(dt, 0, (size_t)n * 4);
  std::vector<int32_t> hi(n);
  for (int i = 0; i < n; ++i) hi[i] = i;
  hipMemcpy(idx, hi.data(), (size_t)n * 4, hipMemcpyHostToDevice);
  WresGeom g;
  wres_plan(3 * D, D, D, D, 1, 1 << 20, &g);
  g.tps = 3; g.n_slices = (D + 31) / 32; g.tail_store = 3; g.gate_stride = D; g.split = 1;
  const int roles = g.n_slices * cells, bps = 64 / roles, nblk = roles * bps * 8, nw = nblk * 4;
  unsigned long long* d;
  hipMalloc(&d, (size_t)nw * 64);
  PanelBatch<Epi> pb;
  const size_t plane = (size_t)cells * n * D;
  for (int i = 0; i < PANEL_MAXP; ++i) {
    const int c = i < cells ? i : 0;
    GruFwdCell cell{i < cells ? n : 0, nullptr, gi + (size_t)c * n * 3 * D, prev + (size_t)c * n * D, idx, dt, nullptr, W + (size_t)c * 3 * D * D, nullptr, bh,
                    H + (size_t)c * n * D, saved + (size_t)c * n * D};
    Epi e{cell, D, plane, 0.1f, nullptr};
    e.dbg = d;
    pb.p[i] = PanelProblem<Epi>{cell.n, cell.prev, cell.prev_idx, cell.w_hh, e};
  }
  hipFuncSetAttribute((const void*)k_gemm_wres<3, Epi, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, WRES_LDS_BYTES);
  hipFuncSetAttribute((const void*)k_gemm_wres<3, Epi, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WRES_LDS_BYTES);
  const size_t lds = (size_t)96 * g.ldk * 4;
  float ms = time_ms([&] { hipLaunchKernelGGL((k_gemm_wres<3, Epi, 0>), dim3(nblk), dim3(256), lds, 0, pb, cells, g, bps); });
  hipMemset(d, 0, (size_t)nw * 64);
  hipLaunchKernelGGL((k_gemm_wres<3, Epi, 8>), dim3(nblk), dim3(256), lds, 0, pb, cells, g, bps);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)nw * 8);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  const char* lab[5] = {"prologue", "panel 1: A latency + MFMAs", "panel 1: gate epilogue", "rest", "whole wave"};
  double sum[5] = {0}, mx[5] = {0};
  int cnt[5] = {0};
  auto add = [&](int k, unsigned long long a, unsigned long long b) { if (a && b && b >= a) { const double v = (double)(b - a); sum[k] += v; if (v > mx[k]) mx[k] = v; cnt[k]++; } };
  for (int w = 0; w < nw; ++w) {
    const unsigned long long* t = &h[(size_t)w * 8];
    add(0, t[0], t[1]); add(1, t[1], t[2]); add(2, t[2], t[3]); add(3, t[3], t[4]); add(4, t[0], t[4]);
  }
  printf("GRU cell level: %d cells x %d rows, D=%d, %d blocks: kernel %.1f us; s_memtime ticks per wave (avg / max)\n", cells, n, D, nblk, ms * 1e3);
  for (int k = 0; k < 5; ++k) printf("   %-32s avg %9.0f  max %9.0f  (n=%d)\n", lab[k], cnt[k] ? sum[k] / cnt[k] : 0.0, mx[k], cnt[k]);
}

int main() {
  const int MM = 120000;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)MM * 600 * 4); hipMalloc(&B, (size_t)600 * 600 * 4); hipMalloc(&C, (size_t)MM * 608 * 4);
  std::vector<float> h((size_t)MM * 600);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)600 * 600 * 4, hipMemcpyHostToDevice);
  gru_timeline(4000, 200);
  timeline<3>("gi-shape 2blk/CU", 8000, 600, 200, A, B, C, 1, 2);
  timeline<3>("gi-shape 1blk/CU", 8000, 600, 200, A, B, C, 1, 1);
  timeline<3>("gi-shape 2blk/CU", 120000, 600, 200, A, B, C, 1, 2);
  for (int M : {0}) {
    run<3, 0>("gi K=200 N=600", M, 600, 200, A, B, C, 1);
    run<2, 0>("gi K=200 N=600", M, 600, 200, A, B, C, 1);
    run<2, 0>("gi K=200 N=600 3blk/CU", M, 600, 200, A, B, C, 1, 3);
    run<1, 0>("gi K=200 N=600 4blk/CU", M, 600, 200, A, B, C, 1, 4);
    run<1, 0>("gi K=200 N=600 6blk/CU", M, 600, 200, A, B, C, 1, 6);
    run<3, 7>("gi mfma only", M, 600, 200, A, B, C, 1);
    run<1, 0>("dprev K=600 N=200", M, 200, 600, A, B, C, 0);
    run<1, 7>("dprev mfma only", M, 200, 600, A, B, C, 0);
    run<3, 0>("loop K=200 N=200", M, 200, 200, A, B, C, 0);
    run<2, 0>("loop K=200 N=200", M, 200, 200, A, B, C, 0);
    run<2, 0>("loop K=200 N=200 3blk/CU", M, 200, 200, A, B, C, 0, 3);
    run<1, 0>("loop K=200 N=200", M, 200, 200, A, B, C, 0);
    run<1, 0>("loop K=200 N=200 4blk/CU", M, 200, 200, A, B, C, 0, 4);
    run<1, 0>("loop K=200 N=200 6blk/CU", M, 200, 200, A, B, C, 0, 6);
  }
  return 0;
}
