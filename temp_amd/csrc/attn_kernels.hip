// Sparse history attention of the self-attention encoder (SARGCNLayer.calc_result / attention,
// reference models/SARGCN.py:25-53), gfx950.
//
// The reference projects a dense (n, T, D) history tensor -- mostly zero rows, masked by -10e9 --
// through k_linear / v_linear for every query row.  Here K/V are projected ONCE per distinct
// (snapshot, node) row into a table; a query row walks only the history positions where its node
// was active (idx >= 0) plus its own current position.  A masked position has softmax weight
// exactly 0 in fp32 (exp(-1e10 - max) == 0 and the current position is never masked), so skipping
// it is exact.
//
// One wave per query row.  Lane l = head * 8 + j (the reference fixes 8 heads); lane (head, j) owns
// columns head*d_k + j + 8*i (i < MAXC) of q / K / V, so a head's dot product is a 3-step xor
// reduction inside its 8-lane group and no LDS is needed.  HBM-bound gather: algorithmic bytes per
// row = (1 + 2 * active positions) * D * 4 read + D * 4 written.
#include "common.hpp"

namespace temp {

struct AttnArgs {
  int n, D, dk, T;
  const float* q; int ldq;
  const float* kh; const float* vh; int ldh;
  const float* kc; const float* vc; int ldc;
  const int32_t* idx;
  const float* decay;
  float sqrt_dk;     // sqrt(d_k): scores are DIVIDED by it like the reference
};

__device__ __forceinline__ float reduce8(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

template <int MAXC>
__global__ __launch_bounds__(256) void k_sa_attn_fwd(AttnArgs a, float* __restrict__ out, float* __restrict__ score,
                                                      float* __restrict__ lse) {
  const int lane = threadIdx.x & 63, head = lane >> 3, j = lane & 7;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n) return;
  int col[MAXC];
  bool ok[MAXC];
  float qv[MAXC], acc[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int d = j + 8 * i;
    ok[i] = d < a.dk;
    col[i] = head * a.dk + (ok[i] ? d : 0);
    qv[i] = ok[i] ? a.q[(size_t)row * a.ldq + col[i]] : 0.f;
    acc[i] = 0.f;
  }
  float m = -INFINITY, ssum = 0.f;
  const int Th = a.T - 1;
  float* srow = score + ((size_t)row * 8 + head) * a.T;
  for (int t = 0; t < a.T; ++t) {
    const float *kp, *vp;
    if (t < Th) {
      const int r = __builtin_amdgcn_readfirstlane(a.idx[(size_t)row * Th + t]);
      if (r < 0) {
        if (j == 0) srow[t] = -INFINITY;
        continue;
      }
      kp = a.kh + (size_t)r * a.ldh;
      vp = a.vh + (size_t)r * a.ldh;
    } else {
      kp = a.kc + (size_t)row * a.ldc;
      vp = a.vc + (size_t)row * a.ldc;
    }
    float kv[MAXC], vv[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      kv[i] = ok[i] ? kp[col[i]] : 0.f;
      vv[i] = ok[i] ? vp[col[i]] : 0.f;
    }
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) part = fmaf(qv[i], kv[i], part);
    float s = reduce8(part) / a.sqrt_dk;
    if (a.decay) s += a.decay[t];
    if (j == 0) srow[t] = s;
    const float mn = fmaxf(m, s);
    const float c = __expf(m - mn), e = __expf(s - mn);
    ssum = ssum * c + e;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) acc[i] = fmaf(e, vv[i], acc[i] * c);
    m = mn;
  }
  const float inv = 1.0f / ssum;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (ok[i]) out[(size_t)row * a.D + (size_t)(j + 8 * i) * 8 + head] = acc[i] * inv;   // reference layout: d*heads + head
  if (j == 0) lse[(size_t)row * 8 + head] = m + __logf(ssum);
}

struct AttnGrads {
  float* d_q; int ld_dq;
  float* d_kh; float* d_vh; int ld_dh;
  float* d_kc; float* d_vc; int ld_dc;
  float* d_decay;
  float* ds;            // [n, 8, T] score gradients for the table pass; when set, no atomics are issued here
};

template <int MAXC>
__global__ __launch_bounds__(256) void k_sa_attn_bwd(AttnArgs a, const float* __restrict__ out, const float* __restrict__ score,
                                                      const float* __restrict__ lse, const float* __restrict__ d_out, AttnGrads g) {
  __shared__ float s_decay[64];
  const bool want_decay = g.d_decay != nullptr;
  if (want_decay) {
    if (threadIdx.x < 64) s_decay[threadIdx.x] = 0.f;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, head = lane >> 3, j = lane & 7;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < a.n) {
    int col[MAXC];
    bool ok[MAXC];
    float qv[MAXC], dov[MAXC], dq[MAXC];
    float dl = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int d = j + 8 * i;
      ok[i] = d < a.dk;
      col[i] = head * a.dk + (ok[i] ? d : 0);
      qv[i] = ok[i] ? a.q[(size_t)row * a.ldq + col[i]] : 0.f;
      const size_t oc = (size_t)row * a.D + (size_t)(ok[i] ? d : 0) * 8 + head;
      dov[i] = ok[i] ? d_out[oc] : 0.f;
      dl = fmaf(dov[i], ok[i] ? out[oc] : 0.f, dl);
      dq[i] = 0.f;
    }
    const float delta = reduce8(dl);
    const float L = lse[(size_t)row * 8 + head];
    const int Th = a.T - 1;
    const float* srow = score + ((size_t)row * 8 + head) * a.T;
    for (int t = 0; t < a.T; ++t) {
      const float *kp, *vp;
      float *dkp, *dvp;
      bool hist = t < Th;
      if (hist) {
        const int r = __builtin_amdgcn_readfirstlane(a.idx[(size_t)row * Th + t]);
        if (r < 0) continue;
        kp = a.kh + (size_t)r * a.ldh;
        vp = a.vh + (size_t)r * a.ldh;
        dkp = g.d_kh + (size_t)r * g.ld_dh;
        dvp = g.d_vh + (size_t)r * g.ld_dh;
      } else {
        kp = a.kc + (size_t)row * a.ldc;
        vp = a.vc + (size_t)row * a.ldc;
        dkp = g.d_kc + (size_t)row * g.ld_dc;
        dvp = g.d_vc + (size_t)row * g.ld_dc;
      }
      float kv[MAXC], vv[MAXC];
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        kv[i] = ok[i] ? kp[col[i]] : 0.f;
        vv[i] = ok[i] ? vp[col[i]] : 0.f;
      }
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) part = fmaf(dov[i], vv[i], part);
      const float dp = reduce8(part);
      const float p = __expf(srow[t] - L);
      const float ds = p * (dp - delta);
      if (want_decay && j == 0) atomicAdd(&s_decay[t], ds);
      if (g.ds && j == 0) g.ds[((size_t)row * 8 + head) * a.T + t] = ds;
      const float gs = ds / a.sqrt_dk;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        dq[i] = fmaf(gs, kv[i], dq[i]);
        if (ok[i]) {
          if (hist) {
            if (!g.ds) {
              atomicAdd(&dkp[col[i]], gs * qv[i]);
              atomicAdd(&dvp[col[i]], p * dov[i]);
            }
          } else {
            dkp[col[i]] = gs * qv[i];
            dvp[col[i]] = p * dov[i];
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (ok[i]) g.d_q[(size_t)row * g.ld_dq + col[i]] = dq[i];
  }
  if (want_decay) {
    __syncthreads();
    if (threadIdx.x < a.T) atomicAdd(&g.d_decay[threadIdx.x], s_decay[threadIdx.x]);
  }
}

// Table pass of the deterministic backward: one wave per history-table row r walks the (query row, position)
// pairs that attended to it (inv_ptr / inv_ref = idx grouped by table row, ref = i * (T-1) + t, built on the
// host) and accumulates  dK[r] = sum ds/sqrt(dk) * q[i],  dV[r] = sum p * dO[i]  in a fixed order -- every table
// row is written exactly once, no atomics, no zero-fill.
template <int MAXC>
__global__ __launch_bounds__(256) void k_sa_attn_bwd_table(AttnArgs a, int n_table, const int32_t* __restrict__ inv_ptr,
                                                            const int32_t* __restrict__ inv_ref, const float* __restrict__ score,
                                                            const float* __restrict__ lse, const float* __restrict__ ds,
                                                            const float* __restrict__ d_out, float* __restrict__ d_kh,
                                                            float* __restrict__ d_vh, int ld_dh) {
  const int lane = threadIdx.x & 63, head = lane >> 3, j = lane & 7;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_table) return;
  int col[MAXC];
  bool ok[MAXC];
  float ak[MAXC], av[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int d = j + 8 * i;
    ok[i] = d < a.dk;
    col[i] = head * a.dk + (ok[i] ? d : 0);
    ak[i] = 0.f; av[i] = 0.f;
  }
  const int Th = a.T - 1;
  const int beg = inv_ptr[r], end = inv_ptr[r + 1];
  for (int e = beg; e < end; ++e) {
    const int ref = __builtin_amdgcn_readfirstlane(inv_ref[e]);
    const int row = ref / Th, t = ref - row * Th;
    const size_t sidx = ((size_t)row * 8 + head) * a.T + t;
    const float p = __expf(score[sidx] - lse[(size_t)row * 8 + head]);
    const float gs = ds[sidx] / a.sqrt_dk;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if (ok[i]) {
        ak[i] = fmaf(gs, a.q[(size_t)row * a.ldq + col[i]], ak[i]);
        av[i] = fmaf(p, d_out[(size_t)row * a.D + (size_t)(j + 8 * i) * 8 + head], av[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (ok[i]) { d_kh[(size_t)r * ld_dh + col[i]] = ak[i]; d_vh[(size_t)r * ld_dh + col[i]] = av[i]; }
}

static int attn_check(const TempAttn* p) {
  if (!p || p->n < 0 || p->D <= 0 || p->T < 1) return TEMP_E_BADARG;
  if (p->heads != 8 || p->D % 8 || p->D / 8 > 64 || p->T > 64) return TEMP_E_UNSUPPORTED;
  if (p->n > 0 && (!p->q || !p->kc || !p->vc)) return TEMP_E_BADARG;
  if (p->n > 0 && p->T > 1 && (!p->idx || !p->kh || !p->vh)) return TEMP_E_BADARG;
  return TEMP_OK;
}

static AttnArgs to_args(const TempAttn* p) {
  AttnArgs a;
  a.n = p->n; a.D = p->D; a.dk = p->D / 8; a.T = p->T;
  a.q = p->q; a.ldq = p->ldq;
  a.kh = p->kh; a.vh = p->vh; a.ldh = p->ldh;
  a.kc = p->kc; a.vc = p->vc; a.ldc = p->ldc;
  a.idx = p->idx; a.decay = p->decay;
  a.sqrt_dk = sqrtf((float)a.dk);
  return a;
}

}  // namespace temp

using namespace temp;

extern "C" {

int temp_sa_attn_fwd(const TempAttn* p, float* out, float* score, float* lse, void* stream) {
  int rc = attn_check(p);
  if (rc != TEMP_OK) return rc;
  if (p->n == 0) return TEMP_OK;
  if (!out || !score || !lse) return TEMP_E_BADARG;
  const AttnArgs a = to_args(p);
  const dim3 grid(ceil_div(p->n, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int mc = (a.dk + 7) / 8;
  if (mc <= 1) TEMP_LAUNCH(K_SA_ATTN_FWD, k_sa_attn_fwd<1>, grid, block, 0, st, a, out, score, lse);
  else if (mc <= 2) TEMP_LAUNCH(K_SA_ATTN_FWD, k_sa_attn_fwd<2>, grid, block, 0, st, a, out, score, lse);
  else if (mc <= 4) TEMP_LAUNCH(K_SA_ATTN_FWD, k_sa_attn_fwd<4>, grid, block, 0, st, a, out, score, lse);
  else TEMP_LAUNCH(K_SA_ATTN_FWD, k_sa_attn_fwd<8>, grid, block, 0, st, a, out, score, lse);
  return launch_status();
}

int temp_sa_attn_bwd(const TempAttn* p, const float* out, const float* score, const float* lse, const float* d_out,
                     float* d_q, int ld_dq, float* d_kh, float* d_vh, int ld_dh, float* d_kc, float* d_vc, int ld_dc,
                     float* d_decay, int n_table, const int32_t* inv_ptr, const int32_t* inv_ref, float* ds_ws, void* stream) {
  int rc = attn_check(p);
  if (rc != TEMP_OK) return rc;
  if (p->n == 0) return TEMP_OK;
  if (!out || !score || !lse || !d_out || !d_q || !d_kc || !d_vc) return TEMP_E_BADARG;
  if (p->T > 1 && (!d_kh || !d_vh)) return TEMP_E_BADARG;
  const bool table_pass = inv_ptr != nullptr && p->T > 1;
  if (table_pass && (!ds_ws || n_table <= 0)) return TEMP_E_BADARG;
  const AttnArgs a = to_args(p);
  AttnGrads g{d_q, ld_dq, d_kh, d_vh, ld_dh, d_kc, d_vc, ld_dc, d_decay, table_pass ? ds_ws : nullptr};
  const dim3 grid(ceil_div(p->n, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int mc = (a.dk + 7) / 8;
  if (mc <= 1) TEMP_LAUNCH(K_SA_ATTN_BWD, k_sa_attn_bwd<1>, grid, block, 0, st, a, out, score, lse, d_out, g);
  else if (mc <= 2) TEMP_LAUNCH(K_SA_ATTN_BWD, k_sa_attn_bwd<2>, grid, block, 0, st, a, out, score, lse, d_out, g);
  else if (mc <= 4) TEMP_LAUNCH(K_SA_ATTN_BWD, k_sa_attn_bwd<4>, grid, block, 0, st, a, out, score, lse, d_out, g);
  else TEMP_LAUNCH(K_SA_ATTN_BWD, k_sa_attn_bwd<8>, grid, block, 0, st, a, out, score, lse, d_out, g);
  if (table_pass) {
    const dim3 tg(ceil_div(n_table, 4));
#define TEMP_TBL(M) TEMP_LAUNCH(K_SA_ATTN_BWD, k_sa_attn_bwd_table<M>, tg, block, 0, st, a, n_table, inv_ptr, inv_ref, score, lse, ds_ws, d_out, d_kh, d_vh, ld_dh)
    if (mc <= 1) TEMP_TBL(1); else if (mc <= 2) TEMP_TBL(2); else if (mc <= 4) TEMP_TBL(4); else TEMP_TBL(8);
#undef TEMP_TBL
  }
  return launch_status();
}

}  // extern "C"
