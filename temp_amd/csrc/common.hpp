// Shared device/host helpers for libtemp_amd (gfx950 / CDNA4 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "temp_amd.h"

#define TEMP_WAVE 64

namespace temp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming (touch-once) accesses: the nt policy keeps them from displacing what other waves re-read from the XCD's L2
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  f32x4 x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 fma4(float s, float4 b, float4 c) {
  return make_float4(fmaf(s, b.x, c.x), fmaf(s, b.y, c.y), fmaf(s, b.z, c.z), fmaf(s, b.w, c.w));
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
__device__ __forceinline__ float4 shfl4(float4 v, int src) {
  return make_float4(__shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src), __shfl(v.w, src));
}
__device__ __forceinline__ float4 shfl_xor4(float4 v, int m) {
  return make_float4(__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m), __shfl_xor(v.w, m));
}
// y = relu(z) was the forward: pass the gradient where y > 0 (the ReLU adjoint folded into the producer of the gradient)
__device__ __forceinline__ float4 relu_gate4(float4 y, float4 g) {
  return make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Persistent, XCD-aware work partition: block b is observed to run on XCD b % 8 (each XCD has its
// own 4 MB L2), so XCD x walks the x-th contiguous eighth of the item list and the rows gathered
// by neighbouring items (same snapshot) stay in one L2.  Only a speed heuristic: any placement
// computes the same result.  Requires gridDim.x % 8 == 0.
struct ItemRange { int beg, end, stride; };
__device__ __forceinline__ ItemRange xcd_items(int n_items, int per_block) {
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int per = (n_items + 7) >> 3;
  ItemRange r;
  const int base = xcd * per;
  r.beg = base + lb * per_block;
  r.end = min(n_items, base + per);
  r.stride = bpx * per_block;
  return r;
}

// The same split for a chunk list whose chunks differ in length by orders of magnitude (ONE large graph with power-law degrees: the
// hubs' full chunks sit at the front of a dst- or src-sorted view -- on the S-hbm shape the first eighth of the chunk list holds
// 57 % of the edges, so with equal chunk COUNTS one XCD worked while seven waited): XCD x owns the chunks whose first edge lies in
// the x-th eighth of the EDGE range.  chunk_beg is non-decreasing (also in device-subsampled views, whose chunks keep their start);
// the two boundaries cost ~20 dependent loads per block, so only lists of >= XCD_BALANCE_MIN chunks take this path.
#define XCD_BALANCE_MIN (1 << 18)
__device__ __forceinline__ int chunk_lower_bound(const int32_t* __restrict__ chunk_beg, int n, int edge) {
  int lo = 0, hi = n;                                         // first chunk with chunk_beg >= edge
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (chunk_beg[mid] < edge) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ void xcd_chunk_range(int n_chunks, int n_edges, const int32_t* __restrict__ chunk_beg, int xcd, int& c_lo, int& c_hi) {
  if (n_chunks < XCD_BALANCE_MIN) {
    const int per = (n_chunks + 7) >> 3;
    c_lo = xcd * per;
    c_hi = min(n_chunks, c_lo + per);
    return;
  }
  const int e_lo = (int)(((long long)n_edges * xcd) >> 3), e_hi = (int)(((long long)n_edges * (xcd + 1)) >> 3);
  c_lo = xcd == 0 ? 0 : chunk_lower_bound(chunk_beg, n_chunks, e_lo);
  c_hi = xcd == 7 ? n_chunks : chunk_lower_bound(chunk_beg, n_chunks, e_hi);
}
__device__ __forceinline__ ItemRange xcd_chunks(int n_chunks, int n_edges, const int32_t* __restrict__ chunk_beg, int per_block) {
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  int c_lo, c_hi;
  xcd_chunk_range(n_chunks, n_edges, chunk_beg, xcd, c_lo, c_hi);
  ItemRange r;
  r.beg = c_lo + lb * per_block;
  r.end = c_hi;
  r.stride = bpx * per_block;
  return r;
}

// Self-loop dropout (include/temp_amd.h: TempDropout): keep-scale of element (row, col), a counter-based hash so the
// backward pass regenerates the forward mask.
struct DropSpec { float p; float inv_keep; unsigned long long seed; };
inline DropSpec drop_spec(const TempDropout* d) {
  DropSpec s{0.f, 1.f, 0ull};
  if (d && d->p > 0.f) { s.p = d->p; s.inv_keep = 1.f / (1.f - d->p); s.seed = d->seed; }
  return s;
}
__device__ __forceinline__ float drop_scale(const DropSpec& d, unsigned row, unsigned col) {
  unsigned long long x = d.seed ^ (((unsigned long long)row << 32) | col);
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  const float u = (float)(x >> 40) * (1.0f / 16777216.0f);
  return u < d.p ? 0.f : d.inv_keep;
}
__device__ __forceinline__ float4 drop4(const DropSpec& d, unsigned row, unsigned col, float4 v) {
  return make_float4(v.x * drop_scale(d, row, col), v.y * drop_scale(d, row, col + 1), v.z * drop_scale(d, row, col + 2),
                     v.w * drop_scale(d, row, col + 3));
}

// process-wide kernel-selection switches (include/temp_amd.h: temp_set_option); definition in gemm_kernels.hip
int option(int key);
void hx_count();                            // diagnostic counter of f16-split kernel launches (temp_f16_launches)

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- optional per-kernel timing (bench only): HIP events recorded on the launch stream around
// every kernel launch between temp_trace_begin() and temp_trace_end().  Off by default; the only
// mutable global of the library, with an explicit caller-driven lifecycle.
enum KernelId {
  K_RGCN_AGG_FWD = 0, K_RGCN_AGG_DX, K_RGCN_DW, K_FIXUP, K_GEMM_LOOP_FWD, K_GEMM_LOOP_DX, K_GEMM_TN, K_REDUCE_SLICES, K_COLSUM,
  K_RELU_BWD, K_GRU_FWD, K_GRU_BWD_GATES, K_GEMM_GRU_DX, K_GEMM_GRU_DPREV, K_GATHER_ROWS, K_SCATTER_ADD, K_DECAY_GRAD, K_COPY,
  K_GEMM_ISO, K_GEMM_GRU_GI, K_GEMM_LINEAR, K_GATHER_CE, K_SA_ATTN_FWD, K_SA_ATTN_BWD, K_GRU_CHAIN_FWD, K_GRU_CHAIN_BWD,
  K_GRU_CHAIN_PACK, K_BX_PACK, K_GEMM_TN_BX8, K_GEMM_TN_BX, K_GRU_WGRAD, K_SEGMENT_SUM, K_KEYS, K_COUNT
};
int trace_open(int kernel_id, hipStream_t st);       // -> slot or -1
void trace_close(int slot, hipStream_t st);
#define TEMP_LAUNCH(KID, kernel, grid, block, shmem, st, ...)            \
  do {                                                                   \
    const int _slot = ::temp::trace_open((KID), (st));                   \
    hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);     \
    ::temp::trace_close(_slot, (st));                                    \
  } while (0)

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? TEMP_OK : TEMP_E_LAUNCH;
}

// --- internal launch helpers shared across translation units (definitions in the .hip files) -----
struct EpiAddBiasAct;   // gemm_kernels.hip

// C[M,N] = act( (row_mask==NULL || row_mask[m] > 0 ? addend[m,n] : 0) + bias[n] + A[M,K] . B )
//   A: row-major, lda; a_idx nullable row gather.  B: [K,N] row-major (ldb) or, if trans_b,
//   stored as [N,K] row-major (ldb).  addend/bias/row_mask nullable.  out may alias addend.
int gemm_add_bias_act(int kid, int M, int N, int K, const float* A, int lda, const int32_t* a_idx,
                      const float* B, int ldb, int trans_b,
                      const float* addend, int ld_add, const int32_t* row_mask, const float* bias, int act,
                      float* out, int ldo, hipStream_t st, const DropSpec* drop = nullptr);   // drop: mask on the PRODUCT (A.B) only

// count <= 4 products  out_i[M_i, N] = bias_i + A_i[M_i, K] . B_i^T  (B_i stored [N, K]) of one shape class in ONE launch (the input
// gates of both directions of the window chain: the second problem's blocks fill the CUs the first one's tail leaves idle)
int gemm_bias_multi(int kid, int count, const int* Ms, int N, int K, const float* const* As, const int32_t* const* a_idxs /*nullable*/, int lda,
                    const float* const* Bs, int ldb,
                    const float* const* biases, float* const* outs, int ldo, hipStream_t st,
                    const unsigned* const* a_keys = nullptr /* per problem: row keys of A by source row (gemm_hx.hpp), nullable */);

// dst[row, :] = src[row, :] * keep-scale(row, col)   (the masked gradient of a dropped-out self-loop message)
int mask_rows(int n, int d, const float* src, float* dst, const DropSpec& drop, hipStream_t st);

// out[Ka,Nb] = sum_m A[m,ka] * B[m,nb]   (split over m, deterministic two-pass reduce)
size_t gemm_tn_workspace(int M, int Ka, int Nb);
bool gemm_tn_can_fuse_bias(int Nb);
int gemm_tn(int M, int Ka, int Nb, const float* A, int lda, const float* B, int ldb,
            float* out, int ldo, void* ws, size_t ws_bytes, hipStream_t st, float* bias_out = nullptr);
// count <= 8 products of one shape class + their bias column sums in one launch and one reduction (gemm_kernels.hip)
bool gemm_tn_multi_bias_supported(int count, int max_m, int Ka, int Nb, int lda, int ldb);
size_t gemm_tn_multi_bias_workspace(int count, int max_m, int Ka, int Nb);
int gemm_tn_multi_bias(int count, const int* Ms, int Ka, int Nb, const float* const* As, int lda, const float* const* Bs, int ldb, float* outs,
                       float* biases, void* ws, size_t ws_bytes, hipStream_t st);

// out = sum over n_slices of part[s] (each `elems` floats, elems % 4 == 0), in slice order
void reduce_slices(int n_slices, size_t elems, int width, const float* part, float* out, int ldo, hipStream_t st, size_t elems2 = 0,
                   const float* part2 = nullptr, float* out2 = nullptr);    // (elems2 > 0: a second, flat array in the same launch)

// out[n_cols] = sum over rows of X[rows, n_cols]
size_t colsum_workspace(int rows, int cols);
int colsum(int rows, int cols, const float* X, int ldx, float* out, void* ws, size_t ws_bytes, hipStream_t st);

// out[s] = sum_{j in [seg_ptr[s], seg_ptr[s+1])} src[order[j]]  (rows with row_mask[row] <= 0 skipped; row_mask nullable)
int segment_sum_rows(int n_seg, int d, const int32_t* seg_ptr, const int32_t* order, const float* src, const int32_t* row_mask, float* out,
                     hipStream_t st, long long n_rows_hint, void* ws = nullptr, size_t ws_bytes = 0, const float* relu_src = nullptr);
// two sources (a masked by mask_a, b unmasked) over the same segmentation; one launch for hot tables, else two single-source calls
int segment_sum_rows2(int n_seg, const int32_t* seg_ptr, const int32_t* order, int d_a, const float* src_a, const int32_t* mask_a, float* out_a,
                      int d_b, const float* src_b, float* out_b, hipStream_t st, long long n_rows_hint);
size_t segment_sum_rows_workspace(int n_seg, long long n_rows, int d);   // n_rows_hint = length of `order` (picks the long-segment kernel)

// dz = (y > 0) ? dy : 0
int relu_bwd(size_t n, const float* y, const float* dy, float* dz, hipStream_t st);

}  // namespace temp
