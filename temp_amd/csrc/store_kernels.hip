// Device-side edge subsample of a resident snapshot (include/temp_amd.h: temp_subsample_views).
//
// In training the target snapshot of every window is message-passed on a uniformly random 50 % of its edges (80 % for
// history snapshots under --random-dropout) with norms recomputed from the subgraph's in-degrees
// (DynamicRGCN.get_batch_graph_embeds models/DynamicRGCN.py:76-90, comp_deg_norm utils/utils.py:74-79).  The reference
// rebuilds a DGL graph on the host for that; here the snapshot's sorted / chunked edge views are already resident in HBM
// (Snapshot.device_views), and the subgraph's views are derived from them in place:
//   k_subsample_select   the exact k-subset: every edge gets the 64-bit key (hash(seed, edge id) << 32 | edge id) -- unique --
//                        and the k-th smallest key is found by an 8-pass radix select (one workgroup per graph, integer
//                        LDS histograms: order-independent, so the draw depends on the seed only);
//   k_subsample_views    one wave per chunk of each of the three views: kept edges are moved to the front of the chunk
//                        (ballot + prefix popcount, order preserved) and chunk_end shrinks.  The chunk table, partial
//                        slots and fix-up lists of the parent stay valid (a chunk may become empty), so every RGCN kernel
//                        runs on the subgraph unchanged; in / out degrees are counted with integer atomics;
//   k_subsample_norm     nnorm = 1 / in_degree (0 for isolated nodes).
#include "common.hpp"

namespace temp {

#define SUB_MAX_JOBS 16

struct SubJob {
  int n_nodes, n_edges, keep;
  unsigned long long seed;
  const int32_t* parent; int32_t* child; const int32_t* eid;          // packs (device), eid [3][E]
  int off_a[3], off_b[3], off_beg[3], off_end[3], off_seg[3], n_chunks[3];
  int off_in_deg, off_out_deg, off_nnorm;
  unsigned char* keep_mask;                                            // nullable [E]
  unsigned long long* thr;                                             // [1] scratch: the k-th smallest key
};
struct SubBatch { SubJob j[SUB_MAX_JOBS]; };

__device__ __forceinline__ unsigned long long sub_key(unsigned long long seed, unsigned e) {
  unsigned long long x = seed ^ (0x9e3779b97f4a7c15ULL * (unsigned long long)(e + 1));
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (x & 0xffffffff00000000ULL) | (unsigned long long)e;
}

__global__ void __launch_bounds__(256) k_subsample_select(SubBatch batch) {
  const SubJob& job = batch.j[blockIdx.x];
  __shared__ int hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining;
  const int E = job.n_edges, k = job.keep;
  if (k <= 0 || E <= 0) { if (threadIdx.x == 0) *job.thr = 0ULL; return; }            // (keep == 0: the views kernel keeps nothing)
  if (k >= E) { if (threadIdx.x == 0) *job.thr = ~0ULL; return; }
  if (threadIdx.x == 0) { s_prefix = 0ULL; s_remaining = k; }
  for (int pass = 0; pass < 8; ++pass) {
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const unsigned long long prefix = s_prefix;
    const unsigned long long mask = pass == 0 ? 0ULL : (~0ULL << (shift + 8));
    for (int e = threadIdx.x; e < E; e += 256) {
      const unsigned long long key = sub_key(job.seed, (unsigned)e);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int rem = s_remaining, b = 0;
      while (b < 255 && hist[b] < rem) { rem -= hist[b]; ++b; }
      s_prefix = prefix | ((unsigned long long)b << shift);
      s_remaining = rem;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *job.thr = s_prefix;                             // keys are unique: exactly k keys are <= it
}

// grid (chunk blocks, 3 views, jobs); 4 waves per block, one chunk per wave-iteration (a chunk has <= 128 edges)
__global__ void __launch_bounds__(256) k_subsample_views(SubBatch batch) {
  const SubJob& job = batch.j[blockIdx.z];
  const int v = blockIdx.y;
  const int nch = job.n_chunks[v];
  const int lane = threadIdx.x & 63;
  const int E = job.n_edges;
  const unsigned long long thr = *job.thr;
  const bool none = job.keep <= 0;
  const int32_t* __restrict__ pa = job.parent + job.off_a[v];
  const int32_t* __restrict__ pb = job.parent + job.off_b[v];
  const int32_t* __restrict__ beg = job.parent + job.off_beg[v];
  const int32_t* __restrict__ end = job.parent + job.off_end[v];
  const int32_t* __restrict__ seg = job.parent + job.off_seg[v];
  const int32_t* __restrict__ eid = job.eid + (size_t)v * E;
  int32_t* __restrict__ ca = job.child + job.off_a[v];
  int32_t* __restrict__ cb = job.child + job.off_b[v];
  int32_t* __restrict__ cend = job.child + job.off_end[v];
  int32_t* deg = v == 0 ? job.child + job.off_in_deg : (v == 1 ? job.child + job.off_out_deg : nullptr);
  for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < nch; c += gridDim.x * 4) {
    const int b0 = beg[c], e0 = end[c];
    int out = b0;
    for (int base = b0; base < e0; base += 64) {                        // <= 2 rounds
      const int p = base + lane;
      bool keep = false;
      int a = 0, b = 0;
      if (p < e0) {
        const int id = eid[p];
        keep = !none && sub_key(job.seed, (unsigned)id) <= thr;
        a = pa[p]; b = pb[p];
        if (v == 0 && job.keep_mask) job.keep_mask[id] = keep ? 1 : 0;
      }
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int pos = out + __popcll(m & ((1ULL << lane) - 1ULL));
        ca[pos] = a; cb[pos] = b;
      }
      out += __popcll(m);
    }
    if (lane == 0) {
      cend[c] = out;
      if (deg && out > b0) atomicAdd(&deg[seg[c]], out - b0);            // integer: order-independent
    }
  }
}

__global__ void __launch_bounds__(256) k_subsample_norm(SubBatch batch) {
  const SubJob& job = batch.j[blockIdx.y];
  const int32_t* deg = job.child + job.off_in_deg;
  float* nn = reinterpret_cast<float*>(job.child + job.off_nnorm);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < job.n_nodes; i += gridDim.x * blockDim.x)
    nn[i] = deg[i] > 0 ? 1.0f / (float)deg[i] : 0.f;                     // comp_deg_norm: 1 / in_deg, inf -> 0
}

}  // namespace temp

using namespace temp;

extern "C" {

int temp_subsample_views(int n_jobs, const TempSubsampleJob* jobs, void* stream) {
  if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return TEMP_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  for (int j0 = 0; j0 < n_jobs; j0 += SUB_MAX_JOBS) {
    const int nj = n_jobs - j0 < SUB_MAX_JOBS ? n_jobs - j0 : SUB_MAX_JOBS;
    SubBatch b = {};
    int max_chunks = 1, max_nodes = 1;
    for (int i = 0; i < nj; ++i) {
      const TempSubsampleJob& s = jobs[j0 + i];
      if (s.n_nodes < 0 || s.n_edges < 0 || s.keep < 0 || s.keep > s.n_edges || !s.parent || !s.child || !s.scratch) return TEMP_E_BADARG;
      if (s.n_edges > 0 && !s.eid) return TEMP_E_BADARG;
      SubJob& d = b.j[i];
      d.n_nodes = s.n_nodes; d.n_edges = s.n_edges; d.keep = s.keep; d.seed = s.seed;
      d.parent = s.parent; d.child = s.child; d.eid = s.eid;
      for (int v = 0; v < 3; ++v) {
        d.off_a[v] = s.off_a[v]; d.off_b[v] = s.off_b[v]; d.off_beg[v] = s.off_chunk_beg[v]; d.off_end[v] = s.off_chunk_end[v];
        d.off_seg[v] = s.off_chunk_seg[v]; d.n_chunks[v] = s.n_chunks[v];
        if (s.n_chunks[v] > max_chunks) max_chunks = s.n_chunks[v];
      }
      d.off_in_deg = s.off_in_deg; d.off_out_deg = s.off_out_deg; d.off_nnorm = s.off_nnorm;
      d.keep_mask = s.keep_mask; d.thr = (unsigned long long*)s.scratch;
      if (s.n_nodes > max_nodes) max_nodes = s.n_nodes;
      // the child's degree counters start from zero (they are accumulated with atomics)
      if (s.n_nodes > 0) {
        if (hipMemsetAsync(s.child + s.off_in_deg, 0, (size_t)s.n_nodes * 4, st) != hipSuccess) return TEMP_E_LAUNCH;
        if (hipMemsetAsync(s.child + s.off_out_deg, 0, (size_t)s.n_nodes * 4, st) != hipSuccess) return TEMP_E_LAUNCH;
      }
    }
    TEMP_LAUNCH(K_COPY, k_subsample_select, dim3(nj), dim3(256), 0, st, b);
    int gx = ceil_div(max_chunks, 4);
    if (gx > 1024) gx = 1024;
    TEMP_LAUNCH(K_COPY, k_subsample_views, dim3(gx, 3, nj), dim3(256), 0, st, b);
    int gn = ceil_div(max_nodes, 256);
    if (gn > 256) gn = 256;
    TEMP_LAUNCH(K_COPY, k_subsample_norm, dim3(gn, nj), dim3(256), 0, st, b);
    const int rc = launch_status();
    if (rc) return rc;
  }
  return TEMP_OK;
}

}  // extern "C"
