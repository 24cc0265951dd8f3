// Probe: k_gru_wgrad (GRU weight gradients from the one gate-gradient matrix, LDS transpose reads) -- semantics of
// ds_read_b64_tr_b16, correctness against fp64, timing, ablations and the shader clock at the headline shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I temp_amd/csrc -I include tools/gru_wgrad_probe.hip -o tools/build/gru_wgrad_probe
#include "common.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
#include <algorithm>
#include <type_traits>
#include "gru_wgrad.hpp"
using namespace temp;
int temp::trace_open(int, hipStream_t) { return -1; }
void temp::trace_close(int, hipStream_t) {}
bx_u32x4* temp::bx_scratch(hipStream_t, size_t) { return nullptr; }
namespace temp { int g_options[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}; }

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

// lane l reads 8 bytes at byte address addr[l] of an LDS image lds[i] = i (16-bit elements)
__global__ void k_tr_probe(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short img[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) img[i] = (short)i;
  __syncthreads();
  const wg_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s4*)((wg_lds_char*)img + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

static void tr_semantics() {
  int h_addr[64];
  // lane l: group g = l >> 4, i = l & 15 -> address of row (i >> 2), columns 4 (i & 3) of a block whose rows are 200 bytes apart,
  // block g starts at 1000 * g elements
  for (int l = 0; l < 64; ++l) { const int g = l >> 4, i = l & 15; h_addr[l] = 2 * (500 * g + (i >> 2) * 100 + 4 * (i & 3)); }
  int* d_addr; short* d_out;
  (void)hipMalloc(&d_addr, sizeof(h_addr)); (void)hipMalloc(&d_out, 64 * 4 * 2);
  (void)hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_tr_probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
  short h_out[256];
  (void)hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int g = l >> 4, i = l & 15;
      const int want = 500 * g + j * 100 + i;                 // row j, column i of the group's block
      if (h_out[l * 4 + j] != want) { if (bad < 8) printf("  tr mismatch lane %d elem %d: got %d want %d\n", l, j, h_out[l * 4 + j], want); ++bad; }
    }
  printf("ds_read_b64_tr_b16 semantics (lane i of a 16-lane group gets column i of the 4 x 16 block whose row r / 4, columns 4 (r %% 4).. lane r points at): %s\n",
         bad ? "DIFFERENT" : "as assumed");
  if (bad) { printf("  lane 0: %d %d %d %d   lane 1: %d %d %d %d   lane 4: %d %d %d %d  lane 16: %d %d %d %d\n", h_out[0], h_out[1], h_out[2], h_out[3], h_out[4], h_out[5], h_out[6], h_out[7],
                    h_out[16], h_out[17], h_out[18], h_out[19], h_out[64], h_out[65], h_out[66], h_out[67]); }
}

[[maybe_unused]] static void split_host(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  unsigned bx; memcpy(&bx, &x, 4);
  const unsigned hb = bx & 0xffff0000u; float hf; memcpy(&hf, &hb, 4);
  const float r1 = x - hf; unsigned b1; memcpy(&b1, &r1, 4);
  const unsigned mb = b1 & 0xffff0000u; float mf; memcpy(&mf, &mb, 4);
  const float r2 = r1 - mf; unsigned b2; memcpy(&b2, &r2, 4);
  h = hb >> 16; m = mb >> 16; l = b2 >> 16;
}

template <int NT>
static void run_case(int count, int M, int d) {
  WgArgs a = {};
  if (!wg_plan(count, d, M, &a)) { printf("plan refused\n"); return; }
  const int Ka = 3 * d;
  const size_t N = (size_t)count * M;
  std::vector<float> g4(N * 4 * d), x(N * d), hd(N * d);
  unsigned st = 777u + M;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : g4) v = rnd() * expf(3.f * rnd());
  for (auto& v : x) v = rnd() * 2.f;
  for (auto& v : hd) v = rnd() * 2.f;
  float *d_x, *d_h, *part, *bpart, *d_g4;
  (void)hipMalloc(&d_g4, g4.size() * 4 + 4096); (void)hipMemcpy(d_g4, g4.data(), g4.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&d_x, x.size() * 4); (void)hipMalloc(&d_h, hd.size() * 4);
  const WgWs ws = wg_workspace(a);
  char* d_ws; float *d_w, *d_b;
  (void)hipMalloc(&d_ws, ws.total); (void)hipMemset(d_ws, 0xff, ws.total);
  (void)hipMalloc(&d_w, (size_t)2 * count * Ka * d * 4); (void)hipMalloc(&d_b, (size_t)2 * count * Ka * 4);
  part = (float*)(d_ws + ws.part); bpart = (float*)(d_ws + ws.bpart);
  a.part2 = (float*)(d_ws + ws.part2); a.bpart2 = (float*)(d_ws + ws.bpart2);
  (void)hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_h, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
  for (int g = 0; g < count; ++g) a.g[g] = WgGroup{M, d_g4 + (size_t)g * M * 4 * d, d_x + (size_t)g * M * d, d_h + (size_t)g * M * d};
  a.part = part; a.bpart = bpart;
  const int grid = 8 * a.per_xcd * a.P;
  const size_t lds = wg_lds_bytes(NT);
  printf("count %d M %d d %d: T %d fb %d r %d mixed %d tail %d P %d per_xcd %d S %d rows/slice %d (tail share %d) grid %d lds %zu\n", count, M, d, a.T, a.fb, a.r, a.mixed,
         a.tail, a.P, a.per_xcd, a.S, a.rows_per_slice, a.rows_per_tail, grid, lds);
#define SETATTR(K) if (hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) printf("setattr failed\n");
  SETATTR((k_gru_wgrad<NT, 0>))
  auto run = [&]() { hipLaunchKernelGGL((k_gru_wgrad<NT, 0>), dim3(grid), dim3(WG_THREADS), lds, 0, a); };
  run();
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return; }
  const float t = time_ms(run);
  const double flops = 2.0 * N * Ka * d * 2;
  printf("  k_gru_wgrad: %.4f ms  = %.1f TFLOP/s algorithmic (%.3f of the bf16/6 roof)\n", t, flops / t / 1e9, flops / t / 1e9 / (2500.0 / 6));
#define ABL(V) { SETATTR((k_gru_wgrad<NT, V>)) auto f = [&]() { hipLaunchKernelGGL((k_gru_wgrad<NT, V>), dim3(grid), dim3(WG_THREADS), lds, 0, a); }; printf("  VAR %3d: %.4f ms\n", V, time_ms(f)); }
  if (M >= 50000) {
    ABL(0) ABL(256) ABL(1) ABL(5) ABL(8) ABL(0) ABL(256)
    run(); (void)hipDeviceSynchronize();
    unsigned long long* dbg; (void)hipMalloc(&dbg, 16 * 8 * 512); a.dbg = dbg;
#define CLK(V) { (void)hipMemset(dbg, 0, 16 * 8 * 512); SETATTR((k_gru_wgrad<NT, V>)) hipLaunchKernelGGL((k_gru_wgrad<NT, V>), dim3(grid), dim3(WG_THREADS), lds, 0, a); (void)hipDeviceSynchronize(); \
      std::vector<unsigned long long> h(2 * 8 * 512); (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost); double c = 0, r = 0; int n = 0; \
      double rmax = 0, rb[8] = {0}; int nb[8] = {0}; \
      for (size_t i = 0; i < h.size(); i += 2) if (h[i + 1]) { c += h[i]; r += h[i + 1]; ++n; rmax = std::max(rmax, (double)h[i + 1]); \
        const int blk = (int)(i / 16), bq = (blk >> 3) % a.P; rb[bq] += h[i + 1]; nb[bq]++; } \
      printf("  VAR %3d: %d waves, mean loop %.0f shader cycles = %.1f us (100 MHz counter) -> %.2f GHz; slowest wave %.1f us; by workgroup of the pair:", V, n, c / n, r / n / 100.0, c / r / 10.0, rmax / 100.0); \
      for (int q = 0; q < a.P; ++q) printf(" %.1f", nb[q] ? rb[q] / nb[q] / 100.0 : 0.0); printf(" us\n"); }
    CLK(64) CLK(320) CLK(69) CLK(65) CLK(64) CLK(320)
    a.dbg = nullptr; (void)hipFree(dbg);
  }
  run(); (void)hipDeviceSynchronize();
  const int R0 = a.tail ? 256 * a.fb : Ka, S2 = a.tail ? a.S * a.P : 0;
  auto reduce = [&]() { hipLaunchKernelGGL(k_gru_wgrad_reduce, dim3(256), dim3(256), 0, 0, 2 * count, Ka, d, a.S, part, bpart, R0, S2, a.part2, a.bpart2, d_w, d_b); };
  reduce();
  printf("  k_gru_wgrad_reduce: %.4f ms\n", time_ms(reduce));
  std::vector<float> hw((size_t)2 * count * Ka * d), hbias((size_t)2 * count * Ka);
  (void)hipMemcpy(hw.data(), d_w, hw.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hbias.data(), d_b, hbias.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, worst_b = 0;
  int checked = 0;
  for (int g = 0; g < count; ++g)
    for (int prod = 0; prod < 2; ++prod)
      for (int ka = 0; ka < Ka; ka += 7) {
        const int acol = (prod == 1 && ka >= 2 * d) ? ka + d : ka;
        const std::vector<float>& B = prod ? hd : x;
        for (int nb = (ka * 3) % 11; nb < d; nb += 37) {
          double ref = 0, mag = 0;
          for (int m = 0; m < M; ++m) {
            const double av = g4[((size_t)g * M + m) * 4 * d + acol], bv = B[((size_t)g * M + m) * d + nb];
            ref += av * bv; mag += fabs(av * bv);
          }
          const double got = hw[((size_t)(2 * g + prod) * Ka + ka) * d + nb];
          worst = std::max(worst, fabs(got - ref) / mag);
          ++checked;
        }
        double ref = 0, mag = 0;
        for (int m = 0; m < M; ++m) { const double av = g4[((size_t)g * M + m) * 4 * d + acol]; ref += av; mag += fabs(av); }
        const double got = hbias[(size_t)(2 * g + prod) * Ka + ka];
        worst_b = std::max(worst_b, fabs(got - ref) / mag);
      }
  printf("  %d sampled outputs: max |err| / sum|a||b| = %.3e   bias sums: %.3e\n", checked, worst, worst_b);
  (void)hipFree(d_g4); (void)hipFree(d_x); (void)hipFree(d_h); (void)hipFree(d_ws); (void)hipFree(d_w); (void)hipFree(d_b);
}

int main() {
  tr_semantics();
  run_case<7>(2, 60000, 200);
  run_case<7>(2, 5003, 200);          // ragged last slab, short slices
  run_case<7>(1, 30000, 200);
  run_case<4>(2, 20000, 104);         // another width: T = 10, mixed workgroup of 2 + 2, even NT (padded row stride)
  run_case<5>(3, 9000, 136);          // T = 13: left-overs 5 + 5 do not share a workgroup
  return 0;
}
