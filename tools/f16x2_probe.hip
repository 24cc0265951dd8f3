// Probe (round 6, verdict item 1): fp32 GEMM through the SCALED TWO-WAY f16 split (3 MFMA products, csrc/split_f16.hpp) against
// the exact three-way bf16 split (6 products, csrc/gemm_bx.hpp) in the same simple panel kernel:
//   * error against fp64, relative to sum |a||b|, on the suite's `_wide` data and on range-stress sets (row magnitudes 2^-30 .. 2^10,
//     column magnitudes 2^-20 .. 2^5, a 2^-100 matrix, within-row spreads of 2^20 and 2^30, GRU-like bounded data with a FIXED scale);
//   * whether the f16 MFMA honours subnormal inputs (the low piece of small elements lives there);
//   * run time of the two arithmetics on the step's shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -I temp_amd/csrc -I include tools/f16x2_probe.hip -o tools/build/f16x2_probe
#include "common.hpp"
#include "split_f16.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
using namespace temp;
int temp::trace_open(int, hipStream_t) { return -1; }
void temp::trace_close(int, hipStream_t) {}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned xb = __float_as_uint(x);
  const unsigned hb = xb & 0xffff0000u;
  const float r1 = x - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mb);
  hi = hb >> 16; mid = mb >> 16; lo = __float_as_uint(r2) >> 16;
}
__device__ __forceinline__ bf16x8_t as_bf(const u32x4 v) { return __builtin_bit_cast(bf16x8_t, v); }

// keys (hx_abs_bits) of the row maxima of X[M][K]
__global__ void __launch_bounds__(256) k_absmax_rows(int M, int K, const float* __restrict__ X, int ldx, unsigned* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  unsigned m = 0;
  for (int k = 4 * lane; k < K; k += 256) m = max(m, hx_abs_bits4(ld4(X + (size_t)row * ldx + k)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if (lane == 0) out[row] = m;
}

#define KC 32
// MODE 0: bf16 x 3 (six products)   MODE 1: f16 x 2 (three products, row keys ra / column keys rb; fixed_a != 0: A scale fixed)
// MODE 2: f16 x 2 + the dropped al.bl product (four products: how much of the error is the dropped term)
// C[M, N] = A[M,K] . Bt^T, Bt = [N][K] row-major
template <int NT, int MODE>
__global__ void __launch_bounds__(256) k_panel(int M, int N, int K, const float* __restrict__ A, int lda, const float* __restrict__ Bt, int ldb,
                                               const unsigned* __restrict__ ra, const unsigned* __restrict__ rb, float fixed_a,
                                               float* __restrict__ out, int ldo) {
  constexpr int BN = NT * 32;
  constexpr int NPL = MODE == 0 ? 3 : 2;
  constexpr int ROWB = KC * 2 + 16;
  constexpr int NV = (BN * KC / 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][NPL][BN * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  const int n0 = blockIdx.y * BN;
  const int arow = m0 + li;
  const bool arow_ok = arow < M;
  const float* aptr = A + (size_t)(arow_ok ? arow : 0) * lda + 8 * hh;
  float sa = 1.f, ia = 1.f;
  if (MODE) {
    const unsigned key = ra[arow_ok ? arow : 0];
    sa = fixed_a != 0.f ? fixed_a : hx_scale(key);
    ia = fixed_a != 0.f ? 1.f / fixed_a : hx_inv_scale(key);
  }
  f32x16 acc[NT];
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float4 breg[NV];
  float bsc[NV];
  auto fetch_b = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int p = threadIdx.x + i * 256;
      const int j = p / (KC / 4), k = (p - j * (KC / 4)) * 4;
      const bool ok = (p < BN * KC / 4) && (n0 + j < N) && (k0 + k < K);
      const float4 v = ld4(Bt + (ok ? (size_t)(n0 + j) * ldb + k0 + k : 0));
      breg[i] = ok ? v : zero4();
      if (MODE) bsc[i] = hx_scale(rb[(n0 + j < N) ? n0 + j : 0]);
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int p = threadIdx.x + i * 256;
      if (p < BN * KC / 4) {
        const int j = p / (KC / 4), k = (p - j * (KC / 4)) * 4;
        const int off = j * ROWB + k * 2;
        if constexpr (MODE == 0) {
          unsigned h0, m0_, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
          split3(breg[i].x, h0, m0_, l0); split3(breg[i].y, h1, m1, l1); split3(breg[i].z, h2, m2, l2); split3(breg[i].w, h3, m3, l3);
          *reinterpret_cast<uint2*>(&Bs[buf][0][off]) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
          *reinterpret_cast<uint2*>(&Bs[buf][1][off]) = make_uint2(m0_ | (m1 << 16), m2 | (m3 << 16));
          *reinterpret_cast<uint2*>(&Bs[buf][2 % NPL][off]) = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
        } else {
          hx_u32x2 H, L;
          hx_split4(breg[i], bsc[i], H, L);
          *reinterpret_cast<uint2*>(&Bs[buf][0][off]) = make_uint2(H[0], H[1]);
          *reinterpret_cast<uint2*>(&Bs[buf][1][off]) = make_uint2(L[0], L[1]);
        }
      }
    }
  };
  constexpr int NS = KC / 16;
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
  const int nchunks = (K + KC - 1) / KC;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) { fetch_b((c + 1) * KC); fetch_a(a0n, a1n, (c + 1) * KC); }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if constexpr (MODE == 0) {
        u32x4 AH, AM, AL;
        const float v[8] = {a0[s].x, a0[s].y, a0[s].z, a0[s].w, a1[s].x, a1[s].y, a1[s].z, a1[s].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned h0, m0_, l0, h1, m1, l1;
          split3(v[2 * i], h0, m0_, l0);
          split3(v[2 * i + 1], h1, m1, l1);
          AH[i] = h0 | (h1 << 16); AM[i] = m0_ | (m1 << 16); AL[i] = l0 | (l1 << 16);
        }
        const bf16x8_t ah = as_bf(AH), am = as_bf(AM), al = as_bf(AL);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int off = (t * 32 + li) * ROWB + (s * 16 + 8 * hh) * 2;
          const bf16x8_t bh = as_bf(*reinterpret_cast<const u32x4*>(&Bs[c & 1][0][off]));
          const bf16x8_t bm = as_bf(*reinterpret_cast<const u32x4*>(&Bs[c & 1][1][off]));
          const bf16x8_t bl = as_bf(*reinterpret_cast<const u32x4*>(&Bs[c & 1][2 % NPL][off]));
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, am, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, ah, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, am, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[t], 0, 0, 0);
        }
      } else {
        hx_u32x4 AH, AL;
        hx_split8(a0[s], a1[s], sa, AH, AL);
        const hx_f16x8 ah = hx_frag(AH), al = hx_frag(AL);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int off = (t * 32 + li) * ROWB + (s * 16 + 8 * hh) * 2;
          const hx_f16x8 bh = hx_frag(*reinterpret_cast<const hx_u32x4*>(&Bs[c & 1][0][off]));
          const hx_f16x8 bl = hx_frag(*reinterpret_cast<const hx_u32x4*>(&Bs[c & 1][1][off]));
          if constexpr (MODE == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, al, acc[t], 0, 0, 0);
          HX_MMA(acc[t], bh, bl, ah, al);
        }
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
      if (col < N) {
        float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        if (MODE) {
          v = scale4(v, ia);
          v.x *= hx_inv_scale(rb[col]); v.y *= hx_inv_scale(rb[col + 1]); v.z *= hx_inv_scale(rb[col + 2]); v.w *= hx_inv_scale(rb[col + 3]);
        }
        st4(out + (size_t)row * ldo + col, v);
      }
    }
}

// one MFMA: a = 2^-20 (an f16 subnormal) in every element, b = 2^10 -> 16 . 2^-10 per output if subnormal inputs are honoured
__global__ void k_denorm(float* out) {
  hx_f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1024.f; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
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

static unsigned long long g_st = 88172645463325252ull;
static double urand() { g_st ^= g_st << 13; g_st ^= g_st >> 7; g_st ^= g_st << 17; return (double)(g_st >> 11) / 9007199254740992.0; }
static double nrand() { const double u = urand() + 1e-300, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }
static float wide(double scale) { return (float)(nrand() * exp(3.0 * urand() - 1.5) * scale); }

struct Set { const char* name; std::vector<float> a, b; float fixed_a; };

int main() {
  const int M = 20000, K = 200, N = 600;
  float* dn; hipMalloc(&dn, 4);
  hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, dn);
  float hd; hipMemcpy(&hd, dn, 4, hipMemcpyDeviceToHost);
  printf("f16 MFMA, subnormal input 2^-20 x 2^10 x 16 terms: got %.9g, exact %.9g -> subnormals %s\n", hd, 16.0 * 1024.0 * 9.5367431640625e-07, hd > 0.f ? "HONOURED" : "FLUSHED");

  std::vector<Set> sets;
  auto fill = [&](const char* name, auto fa, auto fb, float fixed_a = 0.f) {
    Set s; s.name = name; s.a.resize((size_t)M * K); s.b.resize((size_t)N * K); s.fixed_a = fixed_a;
    for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) s.a[(size_t)i * K + k] = fa(i, k);
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) s.b[(size_t)n * K + k] = fb(n, k);
    sets.push_back(std::move(s));
  };
  std::vector<double> rsc(M), csc(N);
  fill("wide (the suite's _wide data)", [&](int, int) { return wide(1.0); }, [&](int, int) { return wide(0.2); });
  for (auto& v : rsc) v = ldexp(1.0, (int)floor(urand() * 41) - 30);
  for (auto& v : csc) v = ldexp(1.0, (int)floor(urand() * 26) - 20);
  fill("range stress: rows 2^-30..2^10, columns 2^-20..2^5", [&](int i, int) { return (float)(wide(1.0) * rsc[i]); }, [&](int n, int) { return (float)(wide(0.2) * csc[n]); });
  fill("tiny: A x 2^-100, B x 2^60", [&](int, int) { return (float)ldexp((double)wide(1.0), -100); }, [&](int, int) { return (float)ldexp((double)wide(0.2), 60); });
  fill("within-row spread 2^20 (every 16th element large)", [&](int, int k) { return (float)(wide(1.0) * ((k & 15) == 3 ? 1048576.0 : 1.0)); }, [&](int, int) { return wide(0.2); });
  fill("within-row spread 2^30 (every 16th element large)", [&](int, int k) { return (float)(wide(1.0) * ((k & 15) == 3 ? 1073741824.0 : 1.0)); }, [&](int, int) { return wide(0.2); });
  fill("GRU-like: a uniform in [-1, 1] with a FIXED scale 2^15, b uniform +-0.07", [&](int, int) { return (float)(2.0 * urand() - 1.0); }, [&](int, int) { return (float)(0.14 * urand() - 0.07); }, 32768.f);
  fill("GRU-like small states: a = 1e-4 x uniform, FIXED scale 2^15", [&](int, int) { return (float)(1e-4 * (2.0 * urand() - 1.0)); }, [&](int, int) { return (float)(0.14 * urand() - 0.07); }, 32768.f);

  float *A, *Bt, *C[3];
  unsigned *ra, *rb;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&Bt, (size_t)N * K * 4);
  for (auto& c : C) hipMalloc(&c, (size_t)M * N * 4);
  hipMalloc(&ra, (size_t)M * 4); hipMalloc(&rb, (size_t)N * 4);
  std::vector<float> c[3];
  for (auto& v : c) v.resize((size_t)M * N);
  printf("\nmax error against fp64 / sum|a||b|  (%d x %d x %d, 1000 sampled rows x all columns)\n", M, K, N);
  printf("%-78s %12s %12s %12s\n", "data", "bf16x3 (6)", "f16x2 (3)", "f16x2+ll (4)");
  for (auto& s : sets) {
    hipMemcpy(A, s.a.data(), s.a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(Bt, s.b.data(), s.b.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_absmax_rows, dim3((M + 3) / 4), dim3(256), 0, 0, M, K, A, K, ra);
    hipLaunchKernelGGL(k_absmax_rows, dim3((N + 3) / 4), dim3(256), 0, 0, N, K, Bt, K, rb);
    const dim3 grid((M + 127) / 128, (N + 127) / 128);
    hipLaunchKernelGGL((k_panel<4, 0>), grid, dim3(256), 0, 0, M, N, K, A, K, Bt, K, ra, rb, s.fixed_a, C[0], N);
    hipLaunchKernelGGL((k_panel<4, 1>), grid, dim3(256), 0, 0, M, N, K, A, K, Bt, K, ra, rb, s.fixed_a, C[1], N);
    hipLaunchKernelGGL((k_panel<4, 2>), grid, dim3(256), 0, 0, M, N, K, A, K, Bt, K, ra, rb, s.fixed_a, C[2], N);
    for (int i = 0; i < 3; ++i) hipMemcpy(c[i].data(), C[i], c[i].size() * 4, hipMemcpyDeviceToHost);
    double e[3] = {0, 0, 0};
    for (int r = 0; r < 1000; ++r) {
      const int row = (int)(((long long)r * 7919) % M);
      for (int n = 0; n < N; ++n) {
        double ref = 0, sabs = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)s.a[(size_t)row * K + k] * (double)s.b[(size_t)n * K + k]; ref += p; sabs += fabs(p); }
        if (sabs == 0) continue;
        for (int i = 0; i < 3; ++i) { const double er = fabs((double)c[i][(size_t)row * N + n] - ref) / sabs; if (!(er <= e[i])) e[i] = er; }
      }
    }
    printf("%-78s %12.3e %12.3e %12.3e\n", s.name, e[0], e[1], e[2]);
  }

  // ---- run time on the step's shapes (same simple kernel, the arithmetic is the only difference)
  printf("\nrun time, simple panel kernel (B slab split in the block, A split in registers)\n");
  struct Shape { int M, K, N, NT; } shapes[] = {{58000, 200, 608, 4}, {58000, 200, 224, 7}, {58000, 600, 224, 7}, {82000, 200, 224, 7}};
  for (auto& sh : shapes) {
    float *a2, *b2, *c2;
    unsigned *r2, *q2;
    hipMalloc(&a2, (size_t)sh.M * sh.K * 4); hipMalloc(&b2, (size_t)sh.N * sh.K * 4); hipMalloc(&c2, (size_t)sh.M * sh.N * 4);
    hipMalloc(&r2, (size_t)sh.M * 4); hipMalloc(&q2, (size_t)sh.N * 4);
    std::vector<float> ha((size_t)sh.M * sh.K), hb((size_t)sh.N * sh.K);
    for (auto& v : ha) v = (float)(2.0 * urand() - 1.0);
    for (auto& v : hb) v = (float)(0.3 * urand() - 0.15);
    hipMemcpy(a2, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b2, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_absmax_rows, dim3((sh.N + 3) / 4), dim3(256), 0, 0, sh.N, sh.K, b2, sh.K, q2);
    const float tmax = time_ms([&]() { hipLaunchKernelGGL(k_absmax_rows, dim3((sh.M + 3) / 4), dim3(256), 0, 0, sh.M, sh.K, a2, sh.K, r2); });
    const double gf = 2.0 * sh.M * sh.K * sh.N / 1e9;
    float t0, t1;
    if (sh.NT == 4) {
      const dim3 grid((sh.M + 127) / 128, (sh.N + 127) / 128);
      t0 = time_ms([&]() { hipLaunchKernelGGL((k_panel<4, 0>), grid, dim3(256), 0, 0, sh.M, sh.N, sh.K, a2, sh.K, b2, sh.K, r2, q2, 0.f, c2, sh.N); });
      t1 = time_ms([&]() { hipLaunchKernelGGL((k_panel<4, 1>), grid, dim3(256), 0, 0, sh.M, sh.N, sh.K, a2, sh.K, b2, sh.K, r2, q2, 0.f, c2, sh.N); });
    } else {
      const dim3 grid((sh.M + 127) / 128, (sh.N + 223) / 224);
      t0 = time_ms([&]() { hipLaunchKernelGGL((k_panel<7, 0>), grid, dim3(256), 0, 0, sh.M, sh.N, sh.K, a2, sh.K, b2, sh.K, r2, q2, 0.f, c2, sh.N); });
      t1 = time_ms([&]() { hipLaunchKernelGGL((k_panel<7, 1>), grid, dim3(256), 0, 0, sh.M, sh.N, sh.K, a2, sh.K, b2, sh.K, r2, q2, 0.f, c2, sh.N); });
    }
    printf("  %6d x %3d x %3d (NT %d): bf16x3 %.1f us = %.0f TF/s   f16x2 %.1f us = %.0f TF/s   (row-maximum pass over A: %.1f us)\n", sh.M, sh.K, sh.N, sh.NT,
           1e3 * t0, gf / t0, 1e3 * t1, gf / t1, 1e3 * tmax);
    hipFree(a2); hipFree(b2); hipFree(c2); hipFree(r2); hipFree(q2);
  }
  return 0;
}
