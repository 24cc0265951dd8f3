// Weights for the f16 two-way split (split_f16.hpp): per-column keys (the column's largest magnitude) and the two f16 planes in
// MFMA fragment order.
//
//   keys[n]                                       = hx_abs_bits of max_k |B[k][n]|   (0 for n >= N up to the tile padding)
//   packed[((s * n_tiles + t) * 2 + p) * 64 + lane] = 16 bytes: plane p (0: h, 1: l) of B[k = 16 s + 8 hh .. + 7][n = 32 t + li] . hx_scale(keys[n])
//                                                   (zero past K / N);  li = lane & 31, hh = lane >> 5
// B is given as [K][N] row-major (trans = 0) or stored as [N][K] (trans = 1), leading dimension ldb.  A product then unscales
// column n of its result by hx_inv_scale(keys[n]).
#pragma once
#include "common.hpp"
#include "split_f16.hpp"

namespace temp {

#define HX_PACK_JOBS 8
struct HxPackJob { const float* B; hx_u32x4* out; unsigned* keys; int K, N, n_tiles, n_slabs, ldb, trans, unit0, kblock0; };
struct HxPackJobs { HxPackJob j[HX_PACK_JOBS]; int count, total_units, total_kblocks; };

inline size_t hx_packed_items(int N, int K) { return (size_t)ceil_div(K, 16) * ceil_div(N, 32) * 128; }   // 16-byte items

// n_slabs > ceil(K / 16): zero slabs behind the matrix (a kernel that walks its slabs in groups)
inline void hx_pack_jobs_add(HxPackJobs& jobs, const float* B, hx_u32x4* out, unsigned* keys, int K, int N, int ldb, int trans, int n_slabs = 0) {
  HxPackJob& j = jobs.j[jobs.count++];
  j.B = B; j.out = out; j.keys = keys; j.K = K; j.N = N; j.n_tiles = ceil_div(N, 32); j.n_slabs = n_slabs > 0 ? n_slabs : ceil_div(K, 16); j.ldb = ldb; j.trans = trans;
  j.unit0 = jobs.total_units; j.kblock0 = jobs.total_kblocks;
  jobs.total_units += j.n_tiles * j.n_slabs;
  jobs.total_kblocks += j.n_tiles;
}

__device__ __forceinline__ const HxPackJob& hx_job_of(const HxPackJobs& jobs, int idx, bool by_kblock, int& local) {
  int ji = 0;
#pragma unroll
  for (int i = 1; i < HX_PACK_JOBS; ++i)
    if (i < jobs.count && idx >= (by_kblock ? jobs.j[i].kblock0 : jobs.j[i].unit0)) ji = i;
  local = idx - (by_kblock ? jobs.j[ji].kblock0 : jobs.j[ji].unit0);
  return jobs.j[ji];
}

// keys AND planes of one 32-column tile of a job in one block (1024 threads): the column maxima first (trans = 1: thread (c, j) reads the float4 pieces j, j + 32, .. of row 32 t + c; trans = 0: thread (g, c)
// reads element (k = g, g + 32, .., column 32 t + c), eight loads in flight: every load instruction reads whole 128-byte lines), then the 16 waves
// deal the tile's slabs among themselves (the two launches cost ~8 us each on an idle chip for a few hundred KB of work)
static __global__ void __launch_bounds__(1024) k_hx_keys_pack(HxPackJobs jobs) {
  __shared__ unsigned sm[32][33];
  __shared__ float scale_l[32];
  int t;
  const HxPackJob& jb = hx_job_of(jobs, blockIdx.x, true, t);
  const int K = jb.K, N = jb.N, ldb = jb.ldb;
  const float* __restrict__ B = jb.B;
  unsigned key = 0;
  int c, g;
  if (jb.trans) {
    c = threadIdx.x >> 5; g = threadIdx.x & 31;
    const int n = 32 * t + c;
    if (n < N) {
      const float* p = B + (size_t)n * ldb;
      if ((K & 3) == 0 && (ldb & 3) == 0) {
        for (int q0 = g; q0 < (K >> 2); q0 += 128) {
          float4 v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { const int q = q0 + 32 * i; v[i] = q < (K >> 2) ? ld4(p + 4 * q) : zero4(); }
#pragma unroll
          for (int i = 0; i < 4; ++i) key = max(key, hx_abs_bits4(v[i]));
        }
      } else {
        for (int k = g; k < K; k += 32) key = max(key, hx_abs_bits(p[k]));
      }
    }
  } else {
    g = threadIdx.x >> 5; c = threadIdx.x & 31;
    const int n = 32 * t + c;
    if (n < N) {
      const float* p = B + n;
      for (int k0 = g; k0 < K; k0 += 256) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int k = k0 + 32 * i; v[i] = k < K ? p[(size_t)k * ldb] : 0.f; }
#pragma unroll
        for (int i = 0; i < 8; ++i) key = max(key, hx_abs_bits(v[i]));
      }
    }
  }
  sm[g][c] = key;
  __syncthreads();
  if (threadIdx.x < 32) {
    key = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) key = max(key, sm[i][threadIdx.x]);
    jb.keys[32 * t + threadIdx.x] = key;
    scale_l[threadIdx.x] = hx_scale(key);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, hh = lane >> 5, li = lane & 31, wave = threadIdx.x >> 6;
  const int n = 32 * t + li;
  const float sc = scale_l[li];
  for (int s0 = wave; s0 < jb.n_slabs; s0 += 16) {
    const int k = 16 * s0 + 8 * hh;
    float4 v0 = zero4(), v1 = zero4();
    if (n < N && k < K) {
      const bool full = k + 4 < K;                           // (K % 4 == 0: the last octet of a K that is no multiple of 8 is half)
      if (jb.trans) {
        const float* p = B + (size_t)n * ldb + k;
        v0 = ld4(p);
        if (full) v1 = ld4(p + 4);
      } else {
        const float* p = B + (size_t)k * ldb + n;
        const size_t l = (size_t)ldb;
        v0 = make_float4(p[0], p[l], p[2 * l], p[3 * l]);
        if (full) v1 = make_float4(p[4 * l], p[5 * l], p[6 * l], p[7 * l]);
      }
    }
    hx_u32x4 H, L;
    hx_split8(v0, v1, sc, H, L);
    hx_u32x4* d = jb.out + ((size_t)s0 * jb.n_tiles + t) * 128 + lane;
    d[0] = H; d[64] = L;
  }
}

inline void hx_pack_launch(const HxPackJobs& jobs, int kid, hipStream_t st) {
  if (!jobs.count) return;
  TEMP_LAUNCH(kid, k_hx_keys_pack, dim3(jobs.total_kblocks), dim3(1024), 0, st, jobs);
}

// all-reduce of an unsigned maximum over the 64 lanes of a wave (every lane active): rows of 16 by DPP, the four rows through
// scalar registers -> a wave-uniform value
__device__ __forceinline__ unsigned hx_wave_max(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));    // quad_perm [1, 0, 3, 2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));    // quad_perm [2, 3, 0, 1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));   // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));   // row_mirror
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}

// out[r][c] = max over the partial rows p < n_part (owned by r: owner == nullptr or owner[owner_stride * p] == r) of part[p][c].
// grid (ceil(cols / 32), n_out), 1024 threads: thread (pg, c) takes p = pg, pg + 32, .. with eight loads in flight, then one LDS pass
// (one thread per column walking all partials was a chain of n_part / 8 L2 round trips: 77 us for 250 x 800).
static __global__ void __launch_bounds__(1024) k_keys_reduce(int n_part, int cols, const unsigned* __restrict__ part, unsigned* __restrict__ out,
                                                             const int32_t* __restrict__ owner, int owner_stride) {
  __shared__ unsigned sm[32][33];
  const int cl = threadIdx.x & 31, pg = threadIdx.x >> 5, c = blockIdx.x * 32 + cl, r = blockIdx.y;
  unsigned key = 0;
  if (c < cols) {
    for (int p0 = pg; p0 < n_part; p0 += 256) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 32 * u;
        v[u] = (p < n_part && (!owner || owner[(size_t)owner_stride * p] == r)) ? part[(size_t)p * cols + c] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) key = max(key, v[u]);
    }
  }
  sm[pg][cl] = key;
  __syncthreads();
  if (threadIdx.x < 32 && blockIdx.x * 32 + threadIdx.x < cols) {
    key = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) key = max(key, sm[i][threadIdx.x]);
    out[(size_t)r * cols + blockIdx.x * 32 + threadIdx.x] = key;
  }
}

// Magnitude keys of a row-major matrix X[n_rows][d] (d % 4 == 0, d <= 256) in one pass over X:
//   row_keys[row] = key of max_c |X[row][c]|                 (nullable)
//   col_part[b][c] = key of max |X[row][c]| over block b's rows (nullable; [gridDim.x][d]) -- k_keys_reduce takes the maxima over the blocks.  (Atomic maxima on
//                    d addresses from every block serialise across the eight L2s: measured 165-315 us for 2 x 30 000 rows.)
// A wave takes whole rows (lane q holds columns 4 q .. 4 q + 3: one coalesced row per load, four rows in flight), the row maximum is
// one DPP reduction, the column maxima stay in the lane until the end.
#define ABSMAX_BLOCKS 1024
// GATHER: row r reads X[idx[r]] (idx < 0: a zero row) and the kernel also WRITES the gathered row to out[r] (temp_gather_rows with
// the keys of its output for free).
// ABSMAX_WAVES waves per block: the column maxima of a block's waves are combined in LDS, so a launch leaves gridDim.x partial
// rows for k_keys_reduce -- with 16-wave blocks a quarter of what 4-wave blocks leave for the same number of waves in flight
// (the reduction over 1 024 partial rows was a 14-us kernel on the step's critical path; over 256 it is launch latency).
#define ABSMAX_WAVES 16
template <bool GATHER>
__global__ void __launch_bounds__(64 * ABSMAX_WAVES) k_absmax_keys(int n_rows, int d, const float* __restrict__ X, int ldx, const int32_t* __restrict__ idx,
                                                     float* __restrict__ out, unsigned* __restrict__ row_keys, unsigned* __restrict__ col_part) {
  __shared__ hx_u32x4 sm[ABSMAX_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d4 = d >> 2;
  const bool act = lane < d4;
  const int gw = blockIdx.x * ABSMAX_WAVES + wave, nw = gridDim.x * ABSMAX_WAVES;
  const float* xp = X + (act ? 4 * lane : 0);
  unsigned ck[4] = {0u, 0u, 0u, 0u};
  for (int r0 = gw; r0 < n_rows; r0 += 4 * nw) {
    float4 v[4];
    int src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int r = r0 + u * nw; const int rr = r < n_rows ? r : r0; src[u] = GATHER ? idx[rr] : rr; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = ld4(xp + (size_t)(src[u] >= 0 ? src[u] : 0) * ldx); if (GATHER && src[u] < 0) v[u] = zero4(); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * nw;
      const unsigned k = act ? hx_abs_bits4(v[u]) : 0u;
      if (r < n_rows) {                                           // (wave-uniform)
        if (GATHER && act) st4(out + (size_t)r * d + 4 * lane, v[u]);
        if (act) { ck[0] = max(ck[0], hx_abs_bits(v[u].x)); ck[1] = max(ck[1], hx_abs_bits(v[u].y)); ck[2] = max(ck[2], hx_abs_bits(v[u].z)); ck[3] = max(ck[3], hx_abs_bits(v[u].w)); }
        if (row_keys) { const unsigned rk = hx_wave_max(k); if (lane == 0) row_keys[r] = rk; }
      }
    }
  }
  if (!col_part) return;
  sm[wave][lane] = hx_u32x4{ck[0], ck[1], ck[2], ck[3]};
  __syncthreads();
  if (wave == 0 && act) {
    hx_u32x4 m = sm[0][lane];
#pragma unroll
    for (int w = 1; w < ABSMAX_WAVES; ++w) { const hx_u32x4 o = sm[w][lane]; m[0] = max(m[0], o[0]); m[1] = max(m[1], o[1]); m[2] = max(m[2], o[2]); m[3] = max(m[3], o[3]); }
    *reinterpret_cast<hx_u32x4*>(col_part + (size_t)blockIdx.x * d + 4 * lane) = m;
  }
}
// (ABSMAX_BLOCKS stays the bound the scratch sizes are computed from -- temp_keys_cols_size; launches use at most 256 blocks of 16 waves)
inline int absmax_blocks(int n_rows) { const int b = ceil_div(n_rows, 8 * ABSMAX_WAVES); return b < 1 ? 1 : (b > ABSMAX_BLOCKS / 4 ? ABSMAX_BLOCKS / 4 : b); }
// col_keys [d] (nullable) needs col_part [absmax_blocks(n_rows)][d] as scratch
static inline void launch_absmax_keys(int n_rows, int d, const float* X, int ldx, unsigned* row_keys, unsigned* col_keys, unsigned* col_part, hipStream_t st) {
  const int blocks = absmax_blocks(n_rows);
  TEMP_LAUNCH(K_KEYS, (k_absmax_keys<false>), dim3(blocks), dim3(64 * ABSMAX_WAVES), 0, st, n_rows, d, X, ldx, (const int32_t*)nullptr, (float*)nullptr, row_keys, col_keys ? col_part : nullptr);
  if (col_keys) TEMP_LAUNCH(K_KEYS, k_keys_reduce, dim3(ceil_div(d, 32), 1), dim3(1024), 0, st, blocks, d, col_part, col_keys, (const int32_t*)nullptr, 0);
}

// out[r] = table[idx[r]] (idx < 0: zero row) with the keys of the OUTPUT: row_keys [n] (nullable), col_keys [d] + col_part scratch (nullable)
static inline void launch_gather_rows_keys(int n, int d, const float* table, const int32_t* idx, float* out, unsigned* row_keys, unsigned* col_keys,
                                           unsigned* col_part, hipStream_t st) {
  const int blocks = absmax_blocks(n);
  TEMP_LAUNCH(K_GATHER_ROWS, (k_absmax_keys<true>), dim3(blocks), dim3(64 * ABSMAX_WAVES), 0, st, n, d, table, d, idx, out, row_keys, col_keys ? col_part : nullptr);
  if (col_keys) TEMP_LAUNCH(K_GATHER_ROWS, k_keys_reduce, dim3(ceil_div(d, 32), 1), dim3(1024), 0, st, blocks, d, col_part, col_keys, (const int32_t*)nullptr, 0);
}

}  // namespace temp
