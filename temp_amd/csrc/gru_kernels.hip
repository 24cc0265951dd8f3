// Recurrent update of the TeMP snapshot encoder on gfx950: decay of the previous node state fused
// with one GRU step (GRRGCNLayer.forward models/RRGCN.py:77-89, BiGRRGCNLayer models/BiRRGCN.py:27-63,
// type-1 cell models/GRU_cell.py:7-31 of the TeMP reference).
//
// Forward is ONE kernel: both gate GEMMs (x . W_ih^T and hdec . W_hh^T) run on fp32 MFMA
// (v_mfma_f32_32x32x2_f32) into four accumulators per 32x32 tile (r, z, i_n, h_n -- r and z share
// one accumulator across both GEMMs), the history gather + exponential decay is applied while the
// A operand is loaded, and sigmoid / tanh / blend run in the epilogue on the accumulators.
// Backward = one pointwise pass for the gate gradients + MFMA GEMMs (gemm_panel.hpp, gemm_tn).
#include "common.hpp"
#include "gemm_panel.hpp"

namespace temp {

#define GRU_KC 40
#define GRU_LDB 33

__device__ __forceinline__ float decay_factor(float dt, float lambda, const float* wb) {
  if (wb) return expf(-fmaxf(fmaf(wb[0], dt, wb[1]), 0.f));
  return expf(-dt * lambda);
}

template <int VARIANT>
__global__ void __launch_bounds__(256) k_gru_fwd(int n, int D, const float* __restrict__ x, const float* __restrict__ prev,
                                                 const int32_t* __restrict__ prev_idx, const float* __restrict__ dt, float lambda,
                                                 const float* __restrict__ decay_wb, const float* __restrict__ w_ih,
                                                 const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                 const float* __restrict__ b_hh, float* __restrict__ h_out, float* __restrict__ saved) {
  __shared__ float Bs[3][GRU_KC * GRU_LDB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  const int j0 = blockIdx.y * 32;
  const int arow = m0 + li;
  const bool arow_ok = arow < n;
  int prow = -1;
  float dec = 0.f;
  if (arow_ok) {
    prow = prev_idx ? prev_idx[arow] : arow;
    dec = decay_factor(dt[arow], lambda, decay_wb);
  }
  f32x16 acc_r, acc_z, acc_in, acc_hn;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_r[r] = 0.f; acc_z[r] = 0.f; acc_in[r] = 0.f; acc_hn[r] = 0.f; }

  // ---- phase X: x . W_ih^T (torch: gates r,z,n; type-1: new gate only) ------------------------
  const float* xa = arow_ok ? x + (size_t)arow * D : nullptr;
  for (int k0 = 0; k0 < D; k0 += GRU_KC) {
    const int kc = min(GRU_KC, D - k0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < 3 * kc * 32; idx += 256) {
      const int g = idx / (kc * 32), rem = idx - g * (kc * 32);
      const int j = rem / kc, k = rem - j * kc;
      float v = 0.f;
      if (j0 + j < D) {
        if (VARIANT == TEMP_GRU_TORCH) v = w_ih[(size_t)(g * D + j0 + j) * D + k0 + k];
        else if (g == 2) v = w_ih[(size_t)(j0 + j) * D + k0 + k];
      }
      Bs[g][k * GRU_LDB + j] = v;
    }
    __syncthreads();
    for (int kk = 0; kk < kc; kk += 8) {
      const int kb = kk + 4 * hh;
      float4 a = zero4();
      if (xa && kb < kc) a = ld4(xa + k0 + kb);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int krow = kb + s;
        const bool kok = krow < kc;
        const int off = krow * GRU_LDB + li;
        if (VARIANT == TEMP_GRU_TORCH) {
          acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], kok ? Bs[0][off] : 0.f, acc_r, 0, 0, 0);
          acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], kok ? Bs[1][off] : 0.f, acc_z, 0, 0, 0);
        }
        acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], kok ? Bs[2][off] : 0.f, acc_in, 0, 0, 0);
      }
    }
  }
  // ---- phase H: hdec . W_hh^T -------------------------------------------------------------------
  const float* ha = (arow_ok && prow >= 0) ? prev + (size_t)prow * D : nullptr;
  for (int k0 = 0; k0 < D; k0 += GRU_KC) {
    const int kc = min(GRU_KC, D - k0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < 3 * kc * 32; idx += 256) {
      const int g = idx / (kc * 32), rem = idx - g * (kc * 32);
      const int j = rem / kc, k = rem - j * kc;
      Bs[g][k * GRU_LDB + j] = (j0 + j < D) ? w_hh[(size_t)(g * D + j0 + j) * D + k0 + k] : 0.f;
    }
    __syncthreads();
    for (int kk = 0; kk < kc; kk += 8) {
      const int kb = kk + 4 * hh;
      float4 a = zero4();
      if (ha && kb < kc) a = scale4(ld4(ha + k0 + kb), dec);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int krow = kb + s;
        const bool kok = krow < kc;
        const int off = krow * GRU_LDB + li;
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], kok ? Bs[0][off] : 0.f, acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], kok ? Bs[1][off] : 0.f, acc_z, 0, 0, 0);
        acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], kok ? Bs[2][off] : 0.f, acc_hn, 0, 0, 0);
      }
    }
  }
  // ---- epilogue -----------------------------------------------------------------------------------
  const int col = j0 + li;
  const bool col_ok = col < D;
  float bir = 0.f, biz = 0.f, bin = 0.f, bhr = 0.f, bhz = 0.f, bhn = 0.f;
  if (col_ok) {
    if (VARIANT == TEMP_GRU_TORCH) { bir = b_ih[col]; biz = b_ih[D + col]; bin = b_ih[2 * D + col]; }
    else bin = b_ih[col];
    bhr = b_hh[col]; bhz = b_hh[D + col]; bhn = b_hh[2 * D + col];
  }
  const size_t nd = (size_t)n * D;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh;   // row inside the tile == the lane that loaded it
    const int prow_r = __shfl(prow, rl);
    const float dec_r = __shfl(dec, rl);
    const int row = m0 + rl;
    if (row < n && col_ok) {
      const float hd = (prow_r >= 0) ? prev[(size_t)prow_r * D + col] * dec_r : 0.f;
      const float rg = 1.f / (1.f + expf(-(acc_r[r] + bir + bhr)));
      const float zg = 1.f / (1.f + expf(-(acc_z[r] + biz + bhz)));
      const float hn = acc_hn[r] + bhn;
      const float ng = tanhf(acc_in[r] + bin + rg * hn);
      const float hy = (VARIANT == TEMP_GRU_TORCH) ? ((1.f - zg) * ng + zg * hd) : (ng + zg * (hd - ng));
      const size_t o = (size_t)row * D + col;
      h_out[o] = hy;
      saved[o] = rg;
      saved[nd + o] = zg;
      saved[2 * nd + o] = ng;
      saved[3 * nd + o] = hn;
      saved[4 * nd + o] = hd;
    }
  }
}

// Gate gradients (pointwise).  dgi: [n, 3D] (torch) or [n, D] (type-1); dgh: [n, 3D]; decv: [n].
template <int VARIANT>
__global__ void __launch_bounds__(256) k_gru_bwd_gates(int n, int D, const float* __restrict__ saved, const float* __restrict__ dh,
                                                       const float* __restrict__ dt, float lambda, const float* __restrict__ decay_wb,
                                                       float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ decv) {
  const size_t nd = (size_t)n * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / D), col = (int)(i - (size_t)row * D);
    const float rg = saved[i], zg = saved[nd + i], ng = saved[2 * nd + i], hn = saved[3 * nd + i], hd = saved[4 * nd + i];
    const float g = dh[i];
    const float dn = g * (1.f - zg);
    const float dz = g * (hd - ng);
    const float dn_pre = dn * (1.f - ng * ng);
    const float dr_pre = dn_pre * hn * rg * (1.f - rg);
    const float dz_pre = dz * zg * (1.f - zg);
    const size_t b3 = (size_t)row * 3 * D + col;
    if (VARIANT == TEMP_GRU_TORCH) {
      dgi[b3] = dr_pre;
      dgi[b3 + D] = dz_pre;
      dgi[b3 + 2 * D] = dn_pre;
    } else {
      dgi[i] = dn_pre;
    }
    dgh[b3] = dr_pre;
    dgh[b3 + D] = dz_pre;
    dgh[b3 + 2 * D] = dn_pre * rg;
    if (col == 0) decv[row] = decay_factor(dt[row], lambda, decay_wb);
  }
}

// d_prev = (dgh . W_hh + dh * z) * decay[row]
struct EpiGruDprev {
  const float* dh; const float* z; const float* decv; float* out; int D;
  __device__ __forceinline__ void operator()(int row, int col, float acc) const {
    const size_t o = (size_t)row * D + col;
    out[o] = (acc + dh[o] * z[o]) * decv[row];
  }
};
struct EpiStore {
  float* out; int ldo;
  __device__ __forceinline__ void operator()(int row, int col, float acc) const { out[(size_t)row * ldo + col] = acc; }
};

// Learnable decay exp(-max(0, w*dt+b)): d/dw = -sum_rows dt*ind*s_row, d/db = -sum_rows ind*s_row with
// s_row = <d_prev[row], prev_row>  (= <d hdec, hdec>).  Single block, fixed order => deterministic.
__global__ void __launch_bounds__(256) k_decay_grad(int n, int D, const float* __restrict__ d_prev, const float* __restrict__ prev,
                                                    const int32_t* __restrict__ prev_idx, const float* __restrict__ dt,
                                                    const float* __restrict__ wb, float* __restrict__ d_wb) {
  __shared__ float sw[256], sb[256];
  float aw = 0.f, ab = 0.f;
  for (int row = threadIdx.x; row < n; row += 256) {
    const int p = prev_idx ? prev_idx[row] : row;
    const float t = dt[row];
    if (p < 0 || fmaf(wb[0], t, wb[1]) <= 0.f) continue;
    float s = 0.f;
    for (int c = 0; c < D; ++c) s = fmaf(d_prev[(size_t)row * D + c], prev[(size_t)p * D + c], s);
    aw -= t * s;
    ab -= s;
  }
  sw[threadIdx.x] = aw;
  sb[threadIdx.x] = ab;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { sw[threadIdx.x] += sw[threadIdx.x + off]; sb[threadIdx.x] += sb[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { d_wb[0] = sw[0]; d_wb[1] = sb[0]; }
}

struct GruBwdWs { float* dgi; float* dgh; float* decv; void* tn; size_t tn_bytes; void* cs; size_t cs_bytes; size_t total; };
static GruBwdWs carve_gru(int n, int d, int variant, char* base) {
  GruBwdWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
  const int gi_w = (variant == TEMP_GRU_TORCH) ? 3 * d : d;
  w.dgi = (float*)take((size_t)n * gi_w * sizeof(float));
  w.dgh = (float*)take((size_t)n * 3 * d * sizeof(float));
  w.decv = (float*)take((size_t)(n > 0 ? n : 1) * sizeof(float));
  w.tn_bytes = gemm_tn_workspace(n, 3 * d, d);
  if (gemm_tn_workspace(n, gi_w, d) > w.tn_bytes) w.tn_bytes = gemm_tn_workspace(n, gi_w, d);
  w.tn = take(w.tn_bytes);
  w.cs_bytes = colsum_workspace(n, 3 * d);
  w.cs = take(w.cs_bytes);
  w.total = off + 256;
  return w;
}

}  // namespace temp

using namespace temp;

extern "C" {

int temp_gru_fwd(int n, int d, int variant, const float* x, const float* prev, const int32_t* prev_idx, const float* dt, float lambda,
                 const float* decay_wb, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* h_out,
                 float* saved, void* stream) {
  if (n < 0 || d <= 0 || !w_ih || !w_hh || !b_ih || !b_hh) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (n > 0 && (!x || !prev || !dt || !h_out || !saved)) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (n == 0) return TEMP_OK;
  dim3 grid(ceil_div(n, 128), ceil_div(d, 32));
  if (variant == TEMP_GRU_TORCH)
    TEMP_LAUNCH(K_GRU_FWD, (k_gru_fwd<TEMP_GRU_TORCH>), grid, dim3(256), 0, (hipStream_t)stream, n, d, x, prev, prev_idx, dt, lambda, decay_wb,
                       w_ih, w_hh, b_ih, b_hh, h_out, saved);
  else
    TEMP_LAUNCH(K_GRU_FWD, (k_gru_fwd<TEMP_GRU_TYPE1>), grid, dim3(256), 0, (hipStream_t)stream, n, d, x, prev, prev_idx, dt, lambda, decay_wb,
                       w_ih, w_hh, b_ih, b_hh, h_out, saved);
  return launch_status();
}

size_t temp_gru_bwd_workspace(int n, int d, int variant) {
  if (n < 0 || d <= 0) return 0;
  return carve_gru(n, d, variant, nullptr).total;
}

int temp_gru_bwd(int n, int d, int variant, const float* x, const float* prev, const int32_t* prev_idx, const float* dt, float lambda,
                 const float* decay_wb, const float* w_ih, const float* w_hh, const float* saved, const float* d_h_out, float* d_x,
                 float* d_prev, float* d_w_ih, float* d_w_hh, float* d_b_ih, float* d_b_hh, float* d_decay_wb, void* workspace,
                 size_t workspace_bytes, void* stream) {
  if (n < 0 || d <= 0 || !w_ih || !w_hh || !d_w_ih || !d_w_hh || !d_b_ih || !d_b_hh) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (n > 0 && (!x || !prev || !dt || !saved || !d_h_out || !d_x || !d_prev)) return TEMP_E_BADARG;
  if (decay_wb && !d_decay_wb) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < temp_gru_bwd_workspace(n, d, variant)) return TEMP_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int gi_w = (variant == TEMP_GRU_TORCH) ? 3 * d : d;
  if (n == 0) {
    if (hipMemsetAsync(d_w_ih, 0, (size_t)gi_w * d * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (hipMemsetAsync(d_w_hh, 0, (size_t)3 * d * d * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (hipMemsetAsync(d_b_ih, 0, (size_t)gi_w * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (hipMemsetAsync(d_b_hh, 0, (size_t)3 * d * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    if (d_decay_wb && hipMemsetAsync(d_decay_wb, 0, 2 * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    return TEMP_OK;
  }
  GruBwdWs w = carve_gru(n, d, variant, (char*)workspace);
  const size_t nd = (size_t)n * d;
  int grid = ceil_div((long long)nd, 256);
  if (grid > 4096) grid = 4096;
  if (variant == TEMP_GRU_TORCH)
    TEMP_LAUNCH(K_GRU_BWD_GATES, (k_gru_bwd_gates<TEMP_GRU_TORCH>), dim3(grid), dim3(256), 0, st, n, d, saved, d_h_out, dt, lambda, decay_wb, w.dgi,
                       w.dgh, w.decv);
  else
    TEMP_LAUNCH(K_GRU_BWD_GATES, (k_gru_bwd_gates<TEMP_GRU_TYPE1>), dim3(grid), dim3(256), 0, st, n, d, saved, d_h_out, dt, lambda, decay_wb, w.dgi,
                       w.dgh, w.decv);
  int rc = launch_status();
  if (rc) return rc;
  // d_x = dgi . W_ih            (W_ih is [gi_w, d] row-major == [K, N])
  rc = launch_gemm_panel(K_GEMM_GRU_DX, n, d, gi_w, w.dgi, gi_w, nullptr, w_ih, d, 0, EpiStore{d_x, d}, st);
  if (rc) return rc;
  // d_prev = (dgh . W_hh + dh * z) * decay
  rc = launch_gemm_panel(K_GEMM_GRU_DPREV, n, d, 3 * d, w.dgh, 3 * d, nullptr, w_hh, d, 0, EpiGruDprev{d_h_out, saved + nd, w.decv, d_prev, d}, st);
  if (rc) return rc;
  // weight / bias gradients
  rc = gemm_tn(n, gi_w, d, w.dgi, gi_w, x, d, d_w_ih, d, w.tn, w.tn_bytes, st);
  if (rc) return rc;
  rc = gemm_tn(n, 3 * d, d, w.dgh, 3 * d, saved + 4 * nd, d, d_w_hh, d, w.tn, w.tn_bytes, st);
  if (rc) return rc;
  rc = colsum(n, gi_w, w.dgi, gi_w, d_b_ih, w.cs, w.cs_bytes, st);
  if (rc) return rc;
  rc = colsum(n, 3 * d, w.dgh, 3 * d, d_b_hh, w.cs, w.cs_bytes, st);
  if (rc) return rc;
  if (decay_wb) {
    TEMP_LAUNCH(K_DECAY_GRAD, k_decay_grad, dim3(1), dim3(256), 0, st, n, d, d_prev, prev, prev_idx, dt, decay_wb, d_decay_wb);
    rc = launch_status();
  }
  return rc;
}

}  // extern "C"
