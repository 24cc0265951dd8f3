// Probe: fp32 GEMM through 3-way bf16 splitting on the bf16 MFMA pipe (6 products) vs the fp32 MFMA panel kernel.
#include "common.hpp"
#include "gemm_wres.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace temp;
int temp::trace_open(int, hipStream_t) { return -1; }
void temp::trace_close(int, hipStream_t) {}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct EpiStoreP {
  float* out; int ldo;
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row_ctx(int) const { return RowCtx(); }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int, int) const { return zero4(); }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4) const { st4(out + (size_t)row * ldo + col, acc); }
};

// x = hi + mid + lo exactly, each piece has <= 8 significant bits (bf16), by truncation
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned xb = __float_as_uint(x);
  const unsigned hb = xb & 0xffff0000u;
  const float r1 = x - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mb);
  hi = hb >> 16; mid = mb >> 16; lo = __float_as_uint(r2) >> 16;
}
// 8 floats -> three fragments of 8 bf16 (4 dwords each)
__device__ __forceinline__ void split8(const float4 a, const float4 b, u32x4& H, u32x4& M, u32x4& L) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3(v[2 * i], h0, m0, l0);
    split3(v[2 * i + 1], h1, m1, l1);
    H[i] = h0 | (h1 << 16); M[i] = m0 | (m1 << 16); L[i] = l0 | (l1 << 16);
  }
}
__device__ __forceinline__ bf16x8_t as_bf(const u32x4 v) { return __builtin_bit_cast(bf16x8_t, v); }

#define BX_KC 32
// C[M, n0..n0+NT*32) = A[M,K] . Bt^T where Bt is [N][K] row-major fp32 (k contiguous)
template <int NT>
__global__ void __launch_bounds__(256) k_panel_bf16x3(int M, int N, int K, const float* __restrict__ A, int lda, const float* __restrict__ Bt,
                                                      int ldb, float* __restrict__ out, int ldo) {
  constexpr int BN = NT * 32;
  constexpr int ROWB = BX_KC * 2 + 16;                     // bytes per (n) row of one piece: 32 bf16 + 16 B pad
  constexpr int NV = (BN * BX_KC / 4 + 255) / 256;          // float4 pieces per thread per chunk
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][3][BN * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  const int n0 = blockIdx.y * BN;
  const int arow = m0 + li;
  const bool arow_ok = arow < M;
  const float* aptr = A + (size_t)(arow_ok ? arow : 0) * lda + 8 * hh;
  f32x16 acc[NT];
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float4 breg[NV];
  auto fetch_b = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int p = threadIdx.x + i * 256;
      const int j = p / (BX_KC / 4), k = (p - j * (BX_KC / 4)) * 4;
      const bool ok = (p < BN * BX_KC / 4) && (n0 + j < N) && (k0 + k < K);
      const float4 v = ld4(Bt + (ok ? (size_t)(n0 + j) * ldb + k0 + k : 0));
      breg[i] = ok ? v : zero4();
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int p = threadIdx.x + i * 256;
      if (p < BN * BX_KC / 4) {
        const int j = p / (BX_KC / 4), k = (p - j * (BX_KC / 4)) * 4;
        unsigned h0, m0_, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
        split3(breg[i].x, h0, m0_, l0); split3(breg[i].y, h1, m1, l1); split3(breg[i].z, h2, m2, l2); split3(breg[i].w, h3, m3, l3);
        const int off = j * ROWB + k * 2;
        *reinterpret_cast<uint2*>(&Bs[buf][0][off]) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
        *reinterpret_cast<uint2*>(&Bs[buf][1][off]) = make_uint2(m0_ | (m1 << 16), m2 | (m3 << 16));
        *reinterpret_cast<uint2*>(&Bs[buf][2][off]) = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
      }
    }
  };
  // A: per 16-wide k slab the lane needs k = k0 + 8*hh .. +7  (two float4)
  constexpr int NS = BX_KC / 16;
  float4 a0[NS], a1[NS], a0n[NS], a1n[NS];
  auto fetch_a = [&](float4 (&x0)[NS], float4 (&x1)[NS], int k0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int kb = k0 + s * 16 + 8 * hh;
      const bool ok0 = arow_ok && (kb < K), ok1 = arow_ok && (kb + 4 < K);
      const float4 v0 = ld4(aptr + (ok0 ? k0 + s * 16 : -8 * hh));
      const float4 v1 = ld4(aptr + (ok1 ? k0 + s * 16 + 4 : -8 * hh));
      x0[s] = ok0 ? v0 : zero4();
      x1[s] = ok1 ? v1 : zero4();
    }
  };
  fetch_b(0);
  fetch_a(a0, a1, 0);
  store_b(0);
  __syncthreads();
  const int nchunks = (K + BX_KC - 1) / BX_KC;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) { fetch_b((c + 1) * BX_KC); fetch_a(a0n, a1n, (c + 1) * BX_KC); }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      u32x4 AH, AM, AL;
      split8(a0[s], a1[s], AH, AM, AL);
      const bf16x8_t ah = as_bf(AH), am = as_bf(AM), al = as_bf(AL);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int off = (t * 32 + li) * ROWB + (s * 16 + 8 * hh) * 2;
        const bf16x8_t bh = as_bf(*reinterpret_cast<const u32x4*>(&Bs[c & 1][0][off]));
        const bf16x8_t bm = as_bf(*reinterpret_cast<const u32x4*>(&Bs[c & 1][1][off]));
        const bf16x8_t bl = as_bf(*reinterpret_cast<const u32x4*>(&Bs[c & 1][2][off]));
        // weights as the A operand (C^T tile, one output row per lane); small terms first
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, am, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, ah, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, am, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[t], 0, 0, 0);
      }
    }
    if (more) {
      store_b((c + 1) & 1);
#pragma unroll
      for (int s = 0; s < NS; ++s) { a0[s] = a0n[s]; a1[s] = a1n[s]; }
    }
    __syncthreads();
  }
  const int row = m0 + li;
  if (row >= M) return;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = n0 + t * 32 + 8 * q + 4 * hh;
      if (col < N) st4(out + (size_t)row * ldo + col, make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]));
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
  float *A, *Bt, *C1, *C2;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&Bt, (size_t)N * K * 4); hipMalloc(&C1, (size_t)M * N * 4); hipMalloc(&C2, (size_t)M * N * 4);
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  unsigned st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : ha) v = rnd() * 2.f;
  for (auto& v : hb) v = rnd() * 0.3f;
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(Bt, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  // fp32 MFMA reference kernel (trans_b = 1: B given as [N][K])
  auto run_f32 = [&]() {
    PanelBatch<EpiStoreP> b;
    for (int i = 0; i < PANEL_MAXP; ++i) b.p[i] = PanelProblem<EpiStoreP>{0, nullptr, nullptr, nullptr, EpiStoreP{C1, N}};
    b.p[0] = PanelProblem<EpiStoreP>{M, A, nullptr, Bt, EpiStoreP{C1, N}};
    hipLaunchKernelGGL((k_gemm_panel<4, EpiStoreP>), dim3((M + 127) / 128, 1, 1), dim3(256), 0, 0, b, N, K, K, K, 1, 0);
    hipLaunchKernelGGL((k_gemm_panel<3, EpiStoreP>), dim3((M + 127) / 128, 1, 1), dim3(256), 0, 0, b, N, K, K, K, 1, 128);
  };
  auto run_bx = [&]() {
    hipLaunchKernelGGL((k_panel_bf16x3<4>), dim3((M + 127) / 128, 2), dim3(256), 0, 0, M, N, K, A, K, Bt, K, C2, N);
  };
  auto run_bx7 = [&]() {
    hipLaunchKernelGGL((k_panel_bf16x3<7>), dim3((M + 127) / 128, 1), dim3(256), 0, 0, M, N, K, A, K, Bt, K, C2, N);
  };
  float t3 = time_ms(run_bx7);
  printf("bf16x3 panel (NT7)       : %.4f ms  %.1f TF/s\n", t3, 2.0 * M * K * N / 1e9 / t3);
  float t1 = time_ms(run_f32), t2 = time_ms(run_bx);
  const double gf = 2.0 * M * K * N / 1e9;
  printf("fp32 MFMA panel (NT4+NT3): %.4f ms  %.1f TF/s\n", t1, gf / t1);
  printf("bf16x3 panel (NT4 x2)    : %.4f ms  %.1f TF/s\n", t2, gf / t2);
  std::vector<float> c1((size_t)M * N), c2((size_t)M * N);
  hipMemcpy(c1.data(), C1, c1.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost);
  double max_rel = 0, max_abs = 0, max_ref_err = 0, max_bx_err = 0;
  for (int r = 0; r < 2000; ++r) {
    const int row = (int)(((long long)r * 7919) % M);
    for (int n = 0; n < N; ++n) {
      double ref = 0, sabs = 0;
      for (int k = 0; k < K; ++k) { ref += (double)ha[(size_t)row * K + k] * hb[(size_t)n * K + k]; sabs += fabs((double)ha[(size_t)row * K + k] * hb[(size_t)n * K + k]); }
      const double e1 = fabs(c1[(size_t)row * N + n] - ref) / sabs, e2 = fabs(c2[(size_t)row * N + n] - ref) / sabs;
      if (e1 > max_ref_err) max_ref_err = e1;
      if (e2 > max_bx_err) max_bx_err = e2;
      const double d = fabs((double)c1[(size_t)row * N + n] - c2[(size_t)row * N + n]);
      if (d > max_abs) max_abs = d;
      if (d / sabs > max_rel) max_rel = d / sabs;
    }
  }
  printf("error vs fp64 / sum|a||b|: fp32-MFMA %.3e   bf16x3 %.3e   (fp32 vs bf16x3: max abs %.3e, rel-to-sum %.3e)\n", max_ref_err, max_bx_err, max_abs, max_rel);
  return 0;
}
