// Grid-barrier latency on MI355X (development probe; sizing of a persistent window-chain kernel, HISTORY.md section 4).
//   hipcc --offload-arch=gfx950 -O3 tools/barrier_probe.hip -o tools/barrier_probe
// A monotonic counter in device memory: every block adds 1 after a release fence and spins (acquire) until the counter reaches
// round * blocks.  Spins are bounded, so a non-co-resident launch reports a timeout instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256, 2) k_barriers(unsigned* ctr, int rounds, unsigned* timeouts, float* sink) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;                    // touch the LDS allocation (occupancy like the GEMM blocks)
  float acc = 0.f;
  for (int r = 1; r <= rounds; ++r) {
    acc += lds[(threadIdx.x + r) & 255];
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)r * gridDim.x;
      long spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > 20000000) { atomicAdd(timeouts, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  if (acc == -1.f) sink[0] = acc;
}

// two-level variant: blocks of one XCD (blockIdx & 7) meet on their own counter; the last one to arrive bumps the global counter
__global__ void __launch_bounds__(256, 2) k_barriers2(unsigned* ctr /* [0] global, [16 + 16 x] per XCD */, int rounds, unsigned* timeouts, float* sink) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;
  float acc = 0.f;
  const int x = blockIdx.x & 7;
  const unsigned per_xcd = (gridDim.x + 7 - x) / 8;           // blocks b with b % 8 == x
  for (int r = 1; r <= rounds; ++r) {
    acc += lds[(threadIdx.x + r) & 255];
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned old = __hip_atomic_fetch_add(ctr + 16 + 16 * x, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == (unsigned)r * per_xcd) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)r * (gridDim.x < 8 ? gridDim.x : 8);
      long spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {      // relaxed polls, ONE acquire fence after
        if (++spins > 20000000) { atomicAdd(timeouts, 1u); break; }
        __builtin_amdgcn_s_sleep(8);
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
  }
  if (acc == -1.f) sink[0] = acc;
}

int main() {
  unsigned *ctr, *to; float* sink;
  hipMalloc(&ctr, 4096); hipMalloc(&to, 4); hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)k_barriers2, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute((const void*)k_barriers, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks : {56, 112, 224, 448, 512}) {
    for (size_t lds : {(size_t)1024, (size_t)78 * 1024}) {
      int occ = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_barriers, 256, lds);
      hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
      if ((long)occ * p.multiProcessorCount < blocks) { printf("blocks %d lds %zu: not co-resident (capacity %d), skipped\n", blocks, lds, occ * p.multiProcessorCount); continue; }
      const int rounds = 200;
      hipMemset(ctr, 0, 4096); hipMemset(to, 0, 4);
      hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(256), lds, 0, ctr, 5, to, sink);       // warm-up
      hipDeviceSynchronize();
      hipMemset(ctr, 0, 4096);
      hipEventRecord(a);
      hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(256), lds, 0, ctr, rounds, to, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms = 0; hipEventElapsedTime(&ms, a, b);
      unsigned h = 0; hipMemcpy(&h, to, 4, hipMemcpyDeviceToHost);
      printf("blocks %3d  lds %5zu KB: %.2f us per grid barrier (%d rounds, %u timeouts)", blocks, lds / 1024, 1e3 * ms / rounds, rounds, h);
      hipMemset(ctr, 0, 4096); hipMemset(to, 0, 4);
      hipLaunchKernelGGL(k_barriers2, dim3(blocks), dim3(256), lds, 0, ctr, 5, to, sink);
      hipDeviceSynchronize();
      hipMemset(ctr, 0, 4096);
      hipEventRecord(a);
      hipLaunchKernelGGL(k_barriers2, dim3(blocks), dim3(256), lds, 0, ctr, rounds, to, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
      hipMemcpy(&h, to, 4, hipMemcpyDeviceToHost);
      printf("   two-level: %.2f us (%u timeouts)\n", 1e3 * ms / rounds, h);
    }
  }
  return 0;
}
