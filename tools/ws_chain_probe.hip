// Feasibility probe (NOT product code): the forward window chain with W_hh STATIONARY in the register files of a cluster of four
// workgroups instead of streamed from L2 per panel-position (HISTORY.md 3c, last paragraph).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/ws_chain_probe tools/ws_chain_probe.hip && gpurun_out/ws_chain_probe
//
// Shape of the headline step: d = 200, 250 panels x 32 tracks x 15 positions, every track active at every position (120 000 rows);
// gi [N, 600] is read, h_out [N, 200] and the five saved planes [5, N, 200] are written -- the same row streams as
// k_gru_chain_fwd.  A cluster = workgroups b, b + 8, b + 16, b + 24 (one XCD under round-robin dispatch; correctness does not
// depend on it).  Member c owns hidden units [50 c, 50 c + 50): their 150 gate columns as five 32-column tiles, one per matrix
// wave, the tile's 13 x 3 split-bf16 B fragments in 156 registers.  A cluster walks its 3-4 panels round-robin; the panel state
// travels between the members once per position as split bf16 planes in A-fragment order through a global exchange buffer,
// published with a device-scope release (fence + flag counter) and consumed with an acquire.  Spins are bounded: a lost hand-over
// ends the kernel with an error flag instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D = 200, G = 600, TR = 32, POS = 15, NPAN = 250, NCL = 64, MEM = 4;
constexpr int UN = D / MEM;             // 50 hidden units per member
constexpr int NT = 5;                   // gate-column tiles per member (150 -> 160)
constexpr int KQ = 13;                  // k-steps of 16 (200 -> 208)
constexpr int PLD = 168;                // LDS row stride of the product buffer (floats)
constexpr int XW = KQ * 3 * 64;         // u32x4 items of one exchange plane set
constexpr int SPIN_MAX = 1 << 22;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split2(float x0, float x1, unsigned& H, unsigned& M, unsigned& L) {
  const unsigned h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
  const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
  const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
  const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
  H = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  M = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
  L = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { const float e = __expf(2.f * x); return 1.f - 2.f / (e + 1.f); }

// wpack: [member][tile][kstep][piece][lane] u32x4;  xchg: [panel][parity][XW] u32x4;  flag: [panel] published member slices
template <int VAR>     // VAR 1: no hand-over waits / fences (timing only, wrong results)   2: no row streams (gi reads, plane writes)
                       //     4: hand-over WITHOUT cache maintenance: exchange data and flags as relaxed agent-scope atomics (per-access
                       //        coherent loads / stores), ordered by s_waitcnt vmcnt(0) on the producer and by program order on the consumer
__global__ void __launch_bounds__(512) k_ws_chain_fwd(const u32x4* __restrict__ wpack, const float* __restrict__ b_hh, const float* __restrict__ gi,
                                                      float* __restrict__ H, float* __restrict__ saved, size_t plane, u32x4* xchg, int* flag, int* err,
                                                      float dec) {
  __shared__ __attribute__((aligned(16))) float P[2][TR * PLD];
  __shared__ __attribute__((aligned(16))) float hown[4][TR * UN];
  __shared__ int cnt;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cl = (b & 7) + 8 * (b >> 5), c = (b >> 3) & 3;
  const int np = (NPAN - cl + NCL - 1) / NCL;               // panels cl, cl + 64, ...
  const int nsteps = np * POS;
  if (tid == 0) cnt = 0;
  __syncthreads();
  if (wave < NT) {
    // ---------------------------------------------------------------- matrix role: tile `wave`
    u32x4 w[KQ][3];
    const u32x4* wp = wpack + ((size_t)(c * NT + wave) * KQ * 3) * 64 + lane;
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int p = 0; p < 3; ++p) w[q][p] = wp[(q * 3 + p) * 64];
    for (int j = 0; j < nsteps; ++j) {
      const int s = j / np, panel = cl + NCL * (j - s * np);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      if (s > 0) {
        if (!(VAR & 1)) {
          int spins = 0;
          if (VAR & 4) {
            while (__hip_atomic_load(flag + panel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < MEM * s) {
              if (++spins > SPIN_MAX) { if (lane == 0) atomicExch(err, 1 + panel); break; }
              __builtin_amdgcn_s_sleep(2);
            }
            asm volatile("" ::: "memory");
          } else {
            while (__hip_atomic_load(flag + panel, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < MEM * s) {
              if (++spins > SPIN_MAX) { if (lane == 0) atomicExch(err, 1 + panel); break; }
              __builtin_amdgcn_s_sleep(2);
            }
          }
        }
        const u32x4* xa = xchg + ((size_t)(panel * 2 + ((s - 1) & 1))) * XW + lane;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          u32x4 xh, xm, xl;
          if (VAR & 4) {
            auto ld = [&](const u32x4* p) {
              unsigned long long* q8 = reinterpret_cast<unsigned long long*>(const_cast<u32x4*>(p));
              const unsigned long long a0 = __hip_atomic_load(q8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const unsigned long long a1 = __hip_atomic_load(q8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              u32x4 v = {(unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32)};
              return v;
            };
            xh = ld(xa + (q * 3) * 64); xm = ld(xa + (q * 3 + 1) * 64); xl = ld(xa + (q * 3 + 2) * 64);
          } else { xh = xa[(q * 3) * 64]; xm = xa[(q * 3 + 1) * 64]; xl = xa[(q * 3 + 2) * 64]; }
          const bf16x8 ah = __builtin_bit_cast(bf16x8, xh), am = __builtin_bit_cast(bf16x8, xm), al = __builtin_bit_cast(bf16x8, xl);
          const bf16x8 wh = __builtin_bit_cast(bf16x8, w[q][0]), wm = __builtin_bit_cast(bf16x8, w[q][1]), wl = __builtin_bit_cast(bf16x8, w[q][2]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc, 0, 0, 0);
        }
      }
      float* pb = P[j & 1] + 32 * wave + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) pb[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * PLD] = acc[r];
      __syncthreads();
    }
    __syncthreads();
  } else {
    // ---------------------------------------------------------------- memory role: gates of 32 tracks x 25 unit pairs
    const int ml = tid - 64 * NT;                             // 0 .. 191
    constexpr int NI = 5, MT = 512 - 64 * NT;
    float2 g[NI][3];
    auto fetch = [&](int j) {
      const int s = j / np, panel = cl + NCL * (j - s * np);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int it = ml + i * MT;
        const int tr = it / 25, u = 2 * (it - tr * 25);
        const size_t row = ((size_t)panel * POS + s) * TR + (it < TR * 25 ? tr : 0);
        const float* src = gi + row * G + UN * c + u;
        if (VAR & 2) { g[i][0] = g[i][1] = g[i][2] = make_float2(0.1f, 0.2f); continue; }
        g[i][0] = *reinterpret_cast<const float2*>(src);
        g[i][1] = *reinterpret_cast<const float2*>(src + D);
        g[i][2] = *reinterpret_cast<const float2*>(src + 2 * D);
      }
    };
    fetch(0);
    __syncthreads();
    for (int j = 0; j < nsteps; ++j) {
      const int s = j / np, slot = j - s * np, panel = cl + NCL * slot;
      float2 gc[NI][3];
#pragma unroll
      for (int i = 0; i < NI; ++i) { gc[i][0] = g[i][0]; gc[i][1] = g[i][1]; gc[i][2] = g[i][2]; }
      if (j + 1 < nsteps) fetch(j + 1);
      const float* pb = P[j & 1];
      unsigned* xo = reinterpret_cast<unsigned*>(xchg + ((size_t)(panel * 2 + (s & 1))) * XW);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int it = ml + i * MT;
        if (it >= TR * 25) continue;
        const int tr = it / 25, u = 2 * (it - tr * 25), U = UN * c + u;
        const float2 ar = *reinterpret_cast<const float2*>(pb + tr * PLD + u), az = *reinterpret_cast<const float2*>(pb + tr * PLD + UN + u),
                     an = *reinterpret_cast<const float2*>(pb + tr * PLD + 2 * UN + u);
        float2 hd = make_float2(0.f, 0.f);
        if (s > 0) { hd = *reinterpret_cast<const float2*>(&hown[slot][tr * UN + u]); hd.x *= dec; hd.y *= dec; }
        const float2 br = *reinterpret_cast<const float2*>(b_hh + U), bz = *reinterpret_cast<const float2*>(b_hh + D + U),
                     bn = *reinterpret_cast<const float2*>(b_hh + 2 * D + U);
        float hv[2], rv[2], zv[2], nv[2], hnv[2];
        const float arv[2] = {ar.x * dec, ar.y * dec}, azv[2] = {az.x * dec, az.y * dec}, anv[2] = {an.x * dec, an.y * dec};
        const float g0[2] = {gc[i][0].x, gc[i][0].y}, g1[2] = {gc[i][1].x, gc[i][1].y}, g2[2] = {gc[i][2].x, gc[i][2].y};
        const float brv[2] = {br.x, br.y}, bzv[2] = {bz.x, bz.y}, bnv[2] = {bn.x, bn.y}, hdv[2] = {hd.x, hd.y};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          rv[k] = sigm(arv[k] + g0[k] + brv[k]);
          zv[k] = sigm(azv[k] + g1[k] + bzv[k]);
          hnv[k] = anv[k] + bnv[k];
          nv[k] = tanh_(g2[k] + rv[k] * hnv[k]);
          hv[k] = (1.f - zv[k]) * nv[k] + zv[k] * hdv[k];
        }
        *reinterpret_cast<float2*>(&hown[slot][tr * UN + u]) = make_float2(hv[0], hv[1]);
        const size_t o = (((size_t)panel * POS + s) * TR + tr) * D + U;
        if (!(VAR & 2)) {
          *reinterpret_cast<float2*>(H + o) = make_float2(hv[0], hv[1]);
          *reinterpret_cast<float2*>(saved + o) = make_float2(rv[0], rv[1]);
          *reinterpret_cast<float2*>(saved + plane + o) = make_float2(zv[0], zv[1]);
          *reinterpret_cast<float2*>(saved + 2 * plane + o) = make_float2(nv[0], nv[1]);
          *reinterpret_cast<float2*>(saved + 3 * plane + o) = make_float2(hnv[0], hnv[1]);
          *reinterpret_cast<float2*>(saved + 4 * plane + o) = make_float2(hdv[0], hdv[1]);
        }
        unsigned hh_, mm_, ll_;
        split2(hv[0], hv[1], hh_, mm_, ll_);
        const int q = U >> 4, half = (U >> 3) & 1, e2 = (U & 7) >> 1;
        unsigned* d0 = xo + ((size_t)(q * 3) * 64 + half * 32 + tr) * 4 + e2;
        if (VAR & 4) {
          __hip_atomic_store(d0, hh_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(d0 + 64 * 4, mm_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(d0 + 2 * 64 * 4, ll_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else { d0[0] = hh_; d0[64 * 4] = mm_; d0[2 * 64 * 4] = ll_; }
      }
      if (VAR & 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's exchange stores are acknowledged at their scope
        int last = 0;
        if (lane == 0) last = (atomicAdd(&cnt, 1) % 3) == 2;
        if (__builtin_amdgcn_readfirstlane(last) && lane == 0) __hip_atomic_fetch_add(flag + panel, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (!(VAR & 1)) {
        __threadfence();                                       // release: this wave's slices are visible device-wide
        int last = 0;
        if (lane == 0) last = (atomicAdd(&cnt, 1) % 3) == 2;
        if (__builtin_amdgcn_readfirstlane(last)) {
          __threadfence();
          if (lane == 0) __hip_atomic_fetch_add(flag + panel, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
    }
  }
}

static void split_host(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  unsigned u; memcpy(&u, &x, 4);
  unsigned hb = u & 0xffff0000u; float hf; memcpy(&hf, &hb, 4);
  float r1 = x - hf; unsigned r1u; memcpy(&r1u, &r1, 4);
  unsigned mb = r1u & 0xffff0000u; float mf; memcpy(&mf, &mb, 4);
  float r2 = r1 - mf; unsigned r2u; memcpy(&r2u, &r2, 4);
  h = (unsigned short)(hb >> 16); m = (unsigned short)(mb >> 16); l = (unsigned short)(r2u >> 16);
}

int main() {
  const size_t N = (size_t)NPAN * POS * TR;
  std::vector<float> W((size_t)G * D), bh(G), gih(N * G);
  unsigned rs = 12345u;
  auto rnd = [&]() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& x : W) x = rnd() * 0.14f;                        // ~U(-1/sqrt(d), 1/sqrt(d))
  for (auto& x : bh) x = rnd() * 0.14f;
  for (auto& x : gih) x = rnd() * 2.f;
  // pack W: fragment lane (n = lane & 31, k = 8 (lane >> 5) + e) of B[k][n] = W_hh[col(n)][k]
  std::vector<unsigned short> wp((size_t)MEM * NT * KQ * 3 * 64 * 8, 0);
  for (int c = 0; c < MEM; ++c) for (int t = 0; t < NT; ++t) for (int q = 0; q < KQ; ++q) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
    const int lc = 32 * t + (l & 31), k = 16 * q + 8 * (l >> 5) + e;
    float v = 0.f;
    if (lc < 3 * UN && k < D) v = W[(size_t)((lc / UN) * D + UN * c + lc % UN) * D + k];
    unsigned short h, m, lo; split_host(v, h, m, lo);
    const size_t base = ((((size_t)(c * NT + t) * KQ + q) * 3) * 64 + l) * 8 + e;
    wp[base] = h; wp[base + 64 * 8] = m; wp[base + 2 * 64 * 8] = lo;
  }
  float *dgi, *dH, *dS, *dbh; u32x4 *dwp, *dx; int *dflag, *derr;
  CK(hipMalloc(&dgi, N * G * 4)); CK(hipMalloc(&dH, N * D * 4)); CK(hipMalloc(&dS, 5 * N * D * 4)); CK(hipMalloc(&dbh, G * 4));
  CK(hipMalloc(&dwp, wp.size() * 2)); CK(hipMalloc(&dx, (size_t)NPAN * 2 * XW * 16)); CK(hipMalloc(&dflag, NPAN * 4)); CK(hipMalloc(&derr, 4));
  CK(hipMemcpy(dgi, gih.data(), N * G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbh, bh.data(), G * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dx, 0, (size_t)NPAN * 2 * XW * 16)); CK(hipMemset(derr, 0, 4));
  const float dec = 0.9f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int var, int reps, const char* what) {
    float best = 1e9f, tot = 0.f;
    for (int i = 0; i < reps + 2; ++i) {
      CK(hipMemsetAsync(dflag, 0, NPAN * 4, 0));
      CK(hipEventRecord(e0, 0));
      if (var == 0) hipLaunchKernelGGL(k_ws_chain_fwd<0>, dim3(4 * NCL), dim3(512), 0, 0, dwp, dbh, dgi, dH, dS, N * D, dx, dflag, derr, dec);
      else if (var == 1) hipLaunchKernelGGL(k_ws_chain_fwd<1>, dim3(4 * NCL), dim3(512), 0, 0, dwp, dbh, dgi, dH, dS, N * D, dx, dflag, derr, dec);
      else if (var == 2) hipLaunchKernelGGL(k_ws_chain_fwd<2>, dim3(4 * NCL), dim3(512), 0, 0, dwp, dbh, dgi, dH, dS, N * D, dx, dflag, derr, dec);
      else if (var == 4) hipLaunchKernelGGL(k_ws_chain_fwd<4>, dim3(4 * NCL), dim3(512), 0, 0, dwp, dbh, dgi, dH, dS, N * D, dx, dflag, derr, dec);
      else if (var == 6) hipLaunchKernelGGL(k_ws_chain_fwd<6>, dim3(4 * NCL), dim3(512), 0, 0, dwp, dbh, dgi, dH, dS, N * D, dx, dflag, derr, dec);
      else hipLaunchKernelGGL(k_ws_chain_fwd<3>, dim3(4 * NCL), dim3(512), 0, 0, dwp, dbh, dgi, dH, dS, N * D, dx, dflag, derr, dec);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { tot += ms; best = ms < best ? ms : best; }
    }
    int err; CK(hipMemcpy(&err, derr, 4, hipMemcpyDeviceToHost));
    printf("%-58s avg %7.1f us  best %7.1f us  (hand-over error flag %d)\n", what, 1e3f * tot / reps, 1e3f * best, err);
  };
  run(0, 20, "weights-stationary forward chain (product)");
  auto check = [&]() {
  // ---- check panels 0, 70, 249 against an fp64 recurrence on the host
  std::vector<float> Hh(N * D);
  CK(hipMemcpy(Hh.data(), dH, N * D * 4, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int panel : {0, 70, 249}) {
    std::vector<double> h(TR * D, 0.0), hn(TR * D);
    for (int s = 0; s < POS; ++s) {
      for (int tr = 0; tr < TR; ++tr) {
        const size_t row = ((size_t)panel * POS + s) * TR + tr;
        for (int u = 0; u < D; ++u) {
          double a[3] = {0, 0, 0};
          if (s > 0) for (int g = 0; g < 3; ++g) { double acc = 0; for (int k = 0; k < D; ++k) acc += h[tr * D + k] * dec * W[(size_t)(g * D + u) * D + k]; a[g] = acc; }
          const double r = 1 / (1 + exp(-(a[0] + gih[row * G + u] + bh[u]))), z = 1 / (1 + exp(-(a[1] + gih[row * G + D + u] + bh[D + u])));
          const double n = tanh(gih[row * G + 2 * D + u] + r * (a[2] + bh[2 * D + u]));
          const double hd = s > 0 ? h[tr * D + u] * dec : 0.0;
          hn[tr * D + u] = (1 - z) * n + z * hd;
          const double err = fabs(hn[tr * D + u] - Hh[row * D + u]);
          worst = err > worst ? err : worst;
        }
      }
      h = hn;
    }
  }
  printf("max |h - fp64 reference| over panels 0, 70, 249, all 15 positions: %.3e  (%s)\n", worst, worst < 2e-5 ? "OK" : "MISMATCH");
  };
  check();
  CK(hipMemset(dH, 0, N * D * 4));
  run(4, 20, "hand-over by per-access coherent atomics, no fences");
  check();
  run(6, 10, "  the same without row streams");
  run(1, 10, "  timing only: no hand-over waits / fences (wrong results)");
  run(2, 10, "  timing only: no row streams (gi reads, plane writes)");
  run(3, 10, "  timing only: neither");
  CK(hipMemset(derr, 0, 4));
  return 0;
}
