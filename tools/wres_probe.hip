// Ablation probe for the weights-resident MFMA GEMM (development tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itemp_amd/csrc tools/wres_probe.hip -o tools/wres_probe
#include "common.hpp"
#include "gemm_wres.hpp"
#include <cstdio>
#include <vector>
using namespace temp;

int temp::trace_open(int, hipStream_t) { return -1; }
void temp::trace_close(int, hipStream_t) {}

struct EpiP {
  float* out; int ldo;
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
  for (int i = 0; i < PANEL_MAXP; ++i) batch.p[i] = PanelProblem<EpiP>{0, nullptr, nullptr, nullptr, EpiP{C, N}};
  batch.p[0] = PanelProblem<EpiP>{M, A, nullptr, B, EpiP{C, N}};
  hipFuncSetAttribute((const void*)k_gemm_wres<NTS, EpiP, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, WRES_LDS_BYTES);
  int bps = 32 * blocks_per_cu / g.n_slices;
  if (bps < 1) bps = 1;
  const size_t lds = (size_t)NTS * 32 * g.ldk * 4;
  float ms = time_ms([&] { hipLaunchKernelGGL((k_gemm_wres<NTS, EpiP, VAR>), dim3(g.n_slices * bps * 8), dim3(256), lds, 0, batch, 1, g, bps); });
  const double fl = 2.0 * M * K * (double)N;
  printf("%-30s M=%6d N=%3d K=%3d NTS=%d slices=%d bps=%2d  %.4f ms  %.1f TF/s useful\n", name, M, N, K, NTS, g.n_slices, bps, ms, fl / ms / 1e9);
}

int main() {
  const int MM = 120000;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)MM * 600 * 4); hipMalloc(&B, (size_t)600 * 600 * 4); hipMalloc(&C, (size_t)MM * 608 * 4);
  std::vector<float> h((size_t)MM * 600);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)600 * 600 * 4, hipMemcpyHostToDevice);
  for (int M : {30000, 120000, 8000}) {
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
