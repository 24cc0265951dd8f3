// Probe: how fast do random row gathers run when the gathered table is HBM-sized, and when a COLUMN SLICE of it fits the 256 MB
// Infinity Cache?  (The HBM-regime window gathers one 800-byte row per edge out of a 419 MB snapshot matrix; a walk over one
// quarter of the columns at a time gathers out of 105 MB.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_probe.hip -o tools/build/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill_idx(int32_t* idx, size_t n, uint32_t rows, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    idx[i] = (int32_t)(h % rows);
  }
}

// one wave per chunk of 64 edges, lane = float4 column of the full row (50 of 64 lanes load)
__global__ void __launch_bounds__(256) k_full(const float4* __restrict__ tab, const int32_t* __restrict__ idx, size_t n_chunks, int D4, float4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t c = wave; c < n_chunks; c += n_waves) {
    const int my = idx[c * 64 + lane];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int col = lane < D4 ? lane : D4 - 1;               // (every lane loads: readlane stays in uniform control flow)
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
      const int r = __builtin_amdgcn_readlane(my, j);
      const float4 v = tab[(size_t)r * D4 + col];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (lane < D4) out[c * D4 + lane] = acc;
  }
}

// a group of G lanes per chunk of 64 edges, lane = float4 column of the slice [c0, c0 + w) of the row
template <int G>
__global__ void __launch_bounds__(256) k_slice(const float4* __restrict__ tab, const int32_t* __restrict__ idx, size_t n_chunks, int D4, int c0, int w,
                                               float4* __restrict__ out) {
  const int lane = threadIdx.x & 63, g = lane / G, l = lane % G;
  constexpr int GPW = 64 / G;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t c = wave * GPW; c < n_chunks; c += n_waves * GPW) {
    const size_t mine = c + g;
    if (mine >= n_chunks) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int32_t* ix = idx + mine * 64;
    if (l < w) {
#pragma unroll 8
      for (int j = 0; j < 64; ++j) {
        const int r = ix[j];
        const float4 v = tab[(size_t)r * D4 + c0 + l];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      out[mine * D4 + c0 + l] = acc;
    }
  }
}

template <class F>
static float time_ms(F f, int iters = 5) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const int D4 = 50;
  const size_t E = (size_t)1 << 24, n_chunks = E / 64;
  int32_t* idx; float4* out;
  CK(hipMalloc(&idx, E * 4));
  CK(hipMalloc(&out, n_chunks * D4 * 16));
  for (int lr = 15; lr <= 21; lr += 2) {
    const uint32_t rows = 1u << lr;
    float4* tab;
    CK(hipMalloc(&tab, (size_t)rows * D4 * 16));
    CK(hipMemset(tab, 0, (size_t)rows * D4 * 16));
    k_fill_idx<<<2048, 256>>>(idx, E, rows, 12345u + lr);
    CK(hipDeviceSynchronize());
    const double gb = (double)E * D4 * 16 / 1e9;
    fflush(stdout); printf("rows 2^%d (table %.0f MB), %zu edges, %.1f GB gathered\n", lr, rows * 800.0 / 1e6, E, gb);
    for (int grid : {2048, 8192}) {
      const float t = time_ms([&] { k_full<<<grid, 256>>>(tab, idx, n_chunks, D4, out); });
      printf("  full rows, grid %5d:            %7.3f ms  %6.2f TB/s\n", grid, t, gb / t);
    }
    for (int ns : {2, 4, 7}) {
      for (int grid : {2048, 8192}) {
        const float t = time_ms([&] {
          for (int s = 0; s < ns; ++s) {
            const int c0 = s * D4 / ns, w = (s + 1) * D4 / ns - c0;
            if (w <= 8) k_slice<8><<<grid, 256>>>(tab, idx, n_chunks, D4, c0, w, out);
            else if (w <= 16) k_slice<16><<<grid, 256>>>(tab, idx, n_chunks, D4, c0, w, out);
            else k_slice<32><<<grid, 256>>>(tab, idx, n_chunks, D4, c0, w, out);
          }
        });
        printf("  %d column slices (%3.0f MB each), grid %5d: %7.3f ms  %6.2f TB/s\n", ns, rows * 800.0 / ns / 1e6, grid, t, gb / t);
      }
    }
    CK(hipFree(tab));
  }
  return 0;
}
