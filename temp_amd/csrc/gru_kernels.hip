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
#include <stdlib.h>
#include "gemm_wres.hpp"
#include "gru_math.hpp"
#include "gru_wgrad_hx.hpp"
#include "side_stream.hpp"

namespace temp {

#define GRU_KC 40
#define GRU_LDB 33

// One kernel, two modes:
//   GI == nullptr : fused -- phase X (x . W_ih^T) and phase H (hdec . W_hh^T) both on MFMA;
//   GI != nullptr : the input-side gates were hoisted out of the recurrence (one big GEMM over all
//                   window positions, temp_gru_input_gates) and only phase H runs here.
// Each phase is a software-pipelined K loop: the three 32-column gate slices of the (transposed)
// weight chunk c+1 and the A fragments of chunk c+1 are fetched into registers (branch-free float4
// loads: out-of-range pieces read a clamped address and are zeroed by a select) while the MFMAs of
// chunk c run out of LDS buffer c&1.
#define GRU_NVB ((3 * GRU_KC * 32 / 4 + 255) / 256)     // float4 weight pieces per thread per chunk (= 4)
#define GRU_NQ (GRU_KC / 8)

// W: [rows, D] row-major; gate g's slice = rows row_base[g] + j0 .. +32 (row_base[g] < 0: gate unused)
struct GruPhase {
  const float* W; int row_base[3]; int D; int j0;
};

__device__ __forceinline__ bool gru_w_piece(int i, const GruPhase& ph, int k0, size_t* off) {
  const int p = threadIdx.x + i * 256;
  const int g = p / (32 * GRU_KC / 4), rem = p - g * (32 * GRU_KC / 4);
  const int j = rem / (GRU_KC / 4), k = (rem - j * (GRU_KC / 4)) * 4;
  const int rb = (g == 0) ? ph.row_base[0] : (g == 1 ? ph.row_base[1] : ph.row_base[2]);
  *off = (size_t)(rb + ph.j0 + j) * ph.D + k0 + k;
  return (p < 3 * 32 * GRU_KC / 4) && rb >= 0 && (ph.j0 + j < ph.D) && (k0 + k < ph.D);
}
// raw loads; the zero-fill select is applied by gru_store_w, after the MFMAs of the current chunk
__device__ __forceinline__ void gru_fetch_w(float4 (&reg)[GRU_NVB], const GruPhase& ph, int k0) {
#pragma unroll
  for (int i = 0; i < GRU_NVB; ++i) {
    size_t off;
    const bool ok = gru_w_piece(i, ph, k0, &off);
    reg[i] = ld4(ph.W + (ok ? off : 0));
  }
}
__device__ __forceinline__ void gru_store_w(const float4 (&reg)[GRU_NVB], float (*Bs)[GRU_KC * GRU_LDB], const GruPhase& ph, int k0) {
#pragma unroll
  for (int i = 0; i < GRU_NVB; ++i) {
    const int p = threadIdx.x + i * 256;
    size_t off;
    const bool ok = gru_w_piece(i, ph, k0, &off);
    const float4 v = ok ? reg[i] : zero4();
    if (p < 3 * 32 * GRU_KC / 4) {
      const int g = p / (32 * GRU_KC / 4), rem = p - g * (32 * GRU_KC / 4);
      const int j = rem / (GRU_KC / 4), k = (rem - j * (GRU_KC / 4)) * 4;
      float* d = Bs[g] + k * GRU_LDB + j;
      d[0] = v.x; d[GRU_LDB] = v.y; d[2 * GRU_LDB] = v.z; d[3 * GRU_LDB] = v.w;
    }
  }
}

// acc0 += A . W_g0^T, acc1 += A . W_g1^T (if G01), acc2 += A . W_g2^T ; A rows scaled by `scale`
template <bool G01>
__device__ __forceinline__ void gru_phase(const GruPhase& ph, const float* __restrict__ arow_ptr, bool a_ok, float scale, int hh, int li,
                                          float (*Bs)[3][GRU_KC * GRU_LDB], f32x16& acc0, f32x16& acc1, f32x16& acc2) {
  const int D = ph.D;
  float4 breg[GRU_NVB], av[GRU_NQ], av_next[GRU_NQ];
  auto fetch_a = [&](float4 (&dst)[GRU_NQ], int k0) {      // raw loads; select_a scales / zero-fills at first use
#pragma unroll
    for (int q = 0; q < GRU_NQ; ++q) {
      const bool ok = a_ok && (k0 + q * 8 + 4 * hh < D);
      dst[q] = ld4(arow_ptr + (ok ? k0 + q * 8 : -4 * hh));
    }
  };
  auto select_a = [&](float4 (&dst)[GRU_NQ], const float4 (&src)[GRU_NQ], int k0) {
#pragma unroll
    for (int q = 0; q < GRU_NQ; ++q) dst[q] = (a_ok && (k0 + q * 8 + 4 * hh < D)) ? scale4(src[q], scale) : zero4();
  };
  __syncthreads();                          // previous phase done with both LDS buffers
  gru_fetch_w(breg, ph, 0);
  fetch_a(av_next, 0);
  gru_store_w(breg, Bs[0], ph, 0);
  select_a(av, av_next, 0);
  __syncthreads();
  const int nchunks = (D + GRU_KC - 1) / GRU_KC;
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    gru_fetch_w(breg, ph, (c + 1) * GRU_KC);      // unconditional (clamped past the end): no value merge, no early wait
    fetch_a(av_next, (c + 1) * GRU_KC);
    __builtin_amdgcn_sched_barrier(0);
    float (*bs)[GRU_KC * GRU_LDB] = Bs[c & 1];
#pragma unroll
    for (int q = 0; q < GRU_NQ; ++q) {
      const float as[4] = {av[q].x, av[q].y, av[q].z, av[q].w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int off = (q * 8 + 4 * hh + s) * GRU_LDB + li;
        if (G01) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bs[0][off], as[s], acc0, 0, 0, 0);      // swapped operands: C^T tile,
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bs[1][off], as[s], acc1, 0, 0, 0);      // one output row per lane
        }
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(bs[2][off], as[s], acc2, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      gru_store_w(breg, Bs[(c + 1) & 1], ph, (c + 1) * GRU_KC);
      select_a(av, av_next, (c + 1) * GRU_KC);
    }
    __syncthreads();
  }
}

#define GRU_MAXP 4
struct GruFwdCell {
  int n; const float* x; const float* gi; const float* prev; const int32_t* prev_idx; const float* dt;
  const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh; float* h_out; float* saved;
};
struct GruFwdBatch { GruFwdCell c[GRU_MAXP]; };

// Cell epilogue shared by the streaming and the weights-resident forward kernels: lane (li, hh) owns row
// `arow` (the row whose A fragment it loaded: prow / dec are its own) and, per q, the 4 consecutive columns
// j0 + 8q + 4hh .. +3 of every gate -> float4 traffic only.
template <int VARIANT, bool HOISTED>
__device__ __forceinline__ void gru_cell_epilogue(const GruFwdCell& cell, int D, size_t plane, int arow, bool arow_ok, int prow, float dec,
                                                  int j0, int hh, const f32x16& acc_r, const f32x16& acc_z, const f32x16& acc_in,
                                                  const f32x16& acc_hn) {
  const float* __restrict__ gi = cell.gi;
  const float* __restrict__ prev = cell.prev;
  const float* __restrict__ b_ih = cell.b_ih;
  const float* __restrict__ b_hh = cell.b_hh;
  float* __restrict__ h_out = cell.h_out;
  float* __restrict__ saved = cell.saved;
  const int G = (VARIANT == TEMP_GRU_TORCH) ? 3 * D : D;
  const int row = arow;
  float4 hd4[4], g0[4], g1[4], g2[4];
  bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {                          // pass 1: all loads in flight
    const int col = j0 + 8 * q + 4 * hh;
    ok[q] = arow_ok && col < D;
    const bool hp = ok[q] && prow >= 0;
    const float4 pv = ld4(prev + (size_t)(hp ? prow : 0) * D + (hp ? col : 0));
    hd4[q] = hp ? scale4(pv, dec) : zero4();
    g0[q] = zero4(); g1[q] = zero4(); g2[q] = zero4();
    if (HOISTED) {
      const float* g = gi + (size_t)(ok[q] ? row : 0) * G + (ok[q] ? col : 0);
      if (VARIANT == TEMP_GRU_TORCH) { g0[q] = ld4(g); g1[q] = ld4(g + D); g2[q] = ld4(g + 2 * D); }
      else g2[q] = ld4(g);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {                          // pass 2: gates, blend, float4 stores
    if (!ok[q]) continue;
    const int col = j0 + 8 * q + 4 * hh;
    float4 bir = zero4(), biz = zero4(), bin = zero4();
    if (!HOISTED) {                                      // hoisted gi already carries b_ih
      if (VARIANT == TEMP_GRU_TORCH) { bir = ld4(b_ih + col); biz = ld4(b_ih + D + col); bin = ld4(b_ih + 2 * D + col); }
      else bin = ld4(b_ih + col);
    }
    const float4 bhr = ld4(b_hh + col), bhz = ld4(b_hh + D + col), bhn = ld4(b_hh + 2 * D + col);
    float o_h[4], o_r[4], o_z[4], o_n[4], o_hn[4];
    const float hdv[4] = {hd4[q].x, hd4[q].y, hd4[q].z, hd4[q].w};
    const float g0v[4] = {g0[q].x, g0[q].y, g0[q].z, g0[q].w}, g1v[4] = {g1[q].x, g1[q].y, g1[q].z, g1[q].w};
    const float g2v[4] = {g2[q].x, g2[q].y, g2[q].z, g2[q].w};
    const float birv[4] = {bir.x, bir.y, bir.z, bir.w}, bizv[4] = {biz.x, biz.y, biz.z, biz.w}, binv[4] = {bin.x, bin.y, bin.z, bin.w};
    const float bhrv[4] = {bhr.x, bhr.y, bhr.z, bhr.w}, bhzv[4] = {bhz.x, bhz.y, bhz.z, bhz.w}, bhnv[4] = {bhn.x, bhn.y, bhn.z, bhn.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      float xr = acc_r[r], xz = acc_z[r], xn = acc_in[r];
      if (HOISTED) {
        if (VARIANT == TEMP_GRU_TORCH) { xr += g0v[e]; xz += g1v[e]; }
        xn = g2v[e];
      }
      const float rg = gate_sigmoid(xr + birv[e] + bhrv[e]);
      const float zg = gate_sigmoid(xz + bizv[e] + bhzv[e]);
      const float hn = acc_hn[r] + bhnv[e];
      const float ng = gate_tanh(xn + binv[e] + rg * hn);
      o_h[e] = (VARIANT == TEMP_GRU_TORCH) ? ((1.f - zg) * ng + zg * hdv[e]) : (ng + zg * (hdv[e] - ng));
      o_r[e] = rg; o_z[e] = zg; o_n[e] = ng; o_hn[e] = hn;
    }
    const size_t o = (size_t)row * D + col;
    st4(h_out + o, make_float4(o_h[0], o_h[1], o_h[2], o_h[3]));
    st4(saved + o, make_float4(o_r[0], o_r[1], o_r[2], o_r[3]));
    st4(saved + plane + o, make_float4(o_z[0], o_z[1], o_z[2], o_z[3]));
    st4(saved + 2 * plane + o, make_float4(o_n[0], o_n[1], o_n[2], o_n[3]));
    st4(saved + 3 * plane + o, make_float4(o_hn[0], o_hn[1], o_hn[2], o_hn[3]));
    st4(saved + 4 * plane + o, hd4[q]);
  }
}

// blockIdx.z selects the cell: the forward- and backward-direction chains advance together.
template <int VARIANT, bool HOISTED>
__global__ void __launch_bounds__(256) k_gru_fwd(GruFwdBatch batch, int D, float lambda, const float* __restrict__ decay_wb, size_t plane) {
  __shared__ float Bs[2][3][GRU_KC * GRU_LDB];
  const GruFwdCell& cell = batch.c[blockIdx.z];
  const int n = cell.n;
  if ((int)blockIdx.x * 128 >= n) return;                     // uniform per block
  const float* __restrict__ x = cell.x;
  const float* __restrict__ gi = cell.gi;
  const float* __restrict__ prev = cell.prev;
  const int32_t* __restrict__ prev_idx = cell.prev_idx;
  const float* __restrict__ dt = cell.dt;
  const float* __restrict__ w_ih = cell.w_ih;
  const float* __restrict__ w_hh = cell.w_hh;
  const float* __restrict__ b_ih = cell.b_ih;
  const float* __restrict__ b_hh = cell.b_hh;
  float* __restrict__ h_out = cell.h_out;
  float* __restrict__ saved = cell.saved;
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

  if (!HOISTED) {                       // phase X: x . W_ih^T (torch: gates r,z,n; type-1: new gate only)
    GruPhase ph;
    ph.W = w_ih; ph.D = D; ph.j0 = j0;
    if (VARIANT == TEMP_GRU_TORCH) { ph.row_base[0] = 0; ph.row_base[1] = D; ph.row_base[2] = 2 * D; }
    else { ph.row_base[0] = -1; ph.row_base[1] = -1; ph.row_base[2] = 0; }
    const float* xa = x + (size_t)(arow_ok ? arow : 0) * D + 4 * hh;
    if (VARIANT == TEMP_GRU_TORCH) gru_phase<true>(ph, xa, arow_ok, 1.f, hh, li, Bs, acc_r, acc_z, acc_in);
    else gru_phase<false>(ph, xa, arow_ok, 1.f, hh, li, Bs, acc_r, acc_z, acc_in);
  }
  {                                     // phase H: hdec . W_hh^T
    GruPhase ph;
    ph.W = w_hh; ph.D = D; ph.j0 = j0;
    ph.row_base[0] = 0; ph.row_base[1] = D; ph.row_base[2] = 2 * D;
    const bool h_ok = arow_ok && prow >= 0;
    const float* ha = prev + (size_t)(h_ok ? prow : 0) * D + 4 * hh;
    gru_phase<true>(ph, ha, h_ok, dec, hh, li, Bs, acc_r, acc_z, acc_hn);
  }
  gru_cell_epilogue<VARIANT, HOISTED>(cell, D, plane, arow, arow_ok, prow, dec, j0, hh, acc_r, acc_z, acc_in, acc_hn);
}

// Grouped epilogue of the weights-resident GEMM (gemm_wres.hpp, GRU mode): the recurrent half of a cell,
//   acc[g] = prev[prev_idx[row]] . W_hh[g]^T   (UNdecayed: the per-row decay commutes with the product),
// finished into gates / new state exactly like gru_cell_epilogue.  Its own operands (hoisted input gates,
// previous-state columns, dt) are loaded while the MFMAs of the panel's last K stage run.
template <int VARIANT>
struct EpiGruCell {
  GruFwdCell cell; int D; size_t plane; float lambda; const float* decay_wb;
  unsigned long long* dbg = nullptr;        // tools/wres_probe.hip only: per-wave s_memtime stamps (VAR bit 3 of the kernel)
  __device__ __forceinline__ void stamp(int k, unsigned long long t) const {
    if ((threadIdx.x & 63) != 0 || !dbg) return;
    unsigned long long* d = dbg + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
    if (d[k] == 0) d[k] = t;
  }
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row_ctx(int) const { return RowCtx(); }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int, int) const { return zero4(); }
  __device__ __forceinline__ void fin4(const RowCtx&, int, int, float4, float4) const {}
  struct GroupPre { float4 hd[4], g0[4], g1[4], g2[4]; float dt; long src; };
  __device__ __forceinline__ GroupPre pre_group(int row, bool row_ok, long a_src, int j0, int hh) const {
    GroupPre p;
    const int G = (VARIANT == TEMP_GRU_TORCH) ? 3 * D : D;
    p.src = row_ok ? a_src : -1;
    p.dt = cell.dt[row_ok ? row : 0];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = j0 + 8 * q + 4 * hh;
      const bool ok = row_ok && col < D;
      const bool hp = ok && a_src >= 0;
      p.hd[q] = ld4(cell.prev + (size_t)(hp ? a_src : 0) * D + (hp ? col : 0));
      const float* g = cell.gi + (size_t)(ok ? row : 0) * G + (ok ? col : 0);
      if (VARIANT == TEMP_GRU_TORCH) { p.g0[q] = ld4(g); p.g1[q] = ld4(g + D); p.g2[q] = ld4(g + 2 * D); }
      else { p.g0[q] = zero4(); p.g1[q] = zero4(); p.g2[q] = ld4(g); }
    }
    return p;
  }
  __device__ __forceinline__ void fin_group(const GroupPre& p, int row, bool row_ok, int j0, int hh, const f32x16 (&acc)[3]) const {
    const float dec = decay_factor(p.dt, lambda, decay_wb);
    float* __restrict__ h_out = cell.h_out;
    float* __restrict__ saved = cell.saved;
    const float* __restrict__ b_hh = cell.b_hh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = j0 + 8 * q + 4 * hh;
      if (!(row_ok && col < D)) continue;
      const bool hp = p.src >= 0;
      const float4 hd4 = hp ? scale4(p.hd[q], dec) : zero4();
      const float4 bhr = ld4(b_hh + col), bhz = ld4(b_hh + D + col), bhn = ld4(b_hh + 2 * D + col);
      const float hdv[4] = {hd4.x, hd4.y, hd4.z, hd4.w};
      const float g0v[4] = {p.g0[q].x, p.g0[q].y, p.g0[q].z, p.g0[q].w}, g1v[4] = {p.g1[q].x, p.g1[q].y, p.g1[q].z, p.g1[q].w};
      const float g2v[4] = {p.g2[q].x, p.g2[q].y, p.g2[q].z, p.g2[q].w};
      const float bhrv[4] = {bhr.x, bhr.y, bhr.z, bhr.w}, bhzv[4] = {bhz.x, bhz.y, bhz.z, bhz.w}, bhnv[4] = {bhn.x, bhn.y, bhn.z, bhn.w};
      float o_h[4], o_r[4], o_z[4], o_n[4], o_hn[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e;
        float xr = acc[0][r] * dec, xz = acc[1][r] * dec;
        if (VARIANT == TEMP_GRU_TORCH) { xr += g0v[e]; xz += g1v[e]; }
        const float rg = gate_sigmoid(xr + bhrv[e]);
        const float zg = gate_sigmoid(xz + bhzv[e]);
        const float hn = acc[2][r] * dec + bhnv[e];
        const float ng = gate_tanh(g2v[e] + rg * hn);
        o_h[e] = (VARIANT == TEMP_GRU_TORCH) ? ((1.f - zg) * ng + zg * hdv[e]) : (ng + zg * (hdv[e] - ng));
        o_r[e] = rg; o_z[e] = zg; o_n[e] = ng; o_hn[e] = hn;
      }
      const size_t o = (size_t)row * D + col;
      st4(h_out + o, make_float4(o_h[0], o_h[1], o_h[2], o_h[3]));
      st4(saved + o, make_float4(o_r[0], o_r[1], o_r[2], o_r[3]));
      st4(saved + plane + o, make_float4(o_z[0], o_z[1], o_z[2], o_z[3]));
      st4(saved + 2 * plane + o, make_float4(o_n[0], o_n[1], o_n[2], o_n[3]));
      st4(saved + 3 * plane + o, make_float4(o_hn[0], o_hn[1], o_hn[2], o_hn[3]));
      st4(saved + 4 * plane + o, hd4);
    }
  }
};

// A cell that starts from the zero state (first position of a chain, the once-per-entity rows of the all-entity pass):
// prev . W_hh^T = 0, so the whole cell is pointwise in the hoisted input gates and b_hh -- no GEMM.  Same outputs as
// EpiGruCell with acc = 0, hdec = 0 (saved planes r, z, n, b_hn, 0).
struct GruZeroCell { int n; const float* gi; const float* b_hh; float* h_out; float* saved; };
struct GruZeroBatch { GruZeroCell c[GRU_MAXP]; };

template <int VARIANT>
__global__ void __launch_bounds__(256) k_gru_fwd_zero(GruZeroBatch batch, int D, size_t plane) {
  const GruZeroCell& cell = batch.c[blockIdx.y];
  const int d4 = D / 4;
  const int G = (VARIANT == TEMP_GRU_TORCH) ? 3 * D : D;
  const size_t total = (size_t)cell.n * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / d4;
    const int col = (int)(i - row * d4) * 4;
    const float* g = cell.gi + row * G + col;
    float4 g0 = zero4(), g1 = zero4(), g2;
    if (VARIANT == TEMP_GRU_TORCH) { g0 = ld4(g); g1 = ld4(g + D); g2 = ld4(g + 2 * D); } else { g2 = ld4(g); }
    const float4 bhr = ld4(cell.b_hh + col), bhz = ld4(cell.b_hh + D + col), bhn = ld4(cell.b_hh + 2 * D + col);
    const float g0v[4] = {g0.x, g0.y, g0.z, g0.w}, g1v[4] = {g1.x, g1.y, g1.z, g1.w}, g2v[4] = {g2.x, g2.y, g2.z, g2.w};
    const float brv[4] = {bhr.x, bhr.y, bhr.z, bhr.w}, bzv[4] = {bhz.x, bhz.y, bhz.z, bhz.w}, bnv[4] = {bhn.x, bhn.y, bhn.z, bhn.w};
    float o_h[4], o_r[4], o_z[4], o_n[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float rg = gate_sigmoid(g0v[e] + brv[e]);
      const float zg = gate_sigmoid(g1v[e] + bzv[e]);
      const float ng = gate_tanh(g2v[e] + rg * bnv[e]);
      o_h[e] = (VARIANT == TEMP_GRU_TORCH) ? (1.f - zg) * ng : (ng - zg * ng);
      o_r[e] = rg; o_z[e] = zg; o_n[e] = ng;
    }
    const size_t o = row * D + col;
    st4(cell.h_out + o, make_float4(o_h[0], o_h[1], o_h[2], o_h[3]));
    st4(cell.saved + o, make_float4(o_r[0], o_r[1], o_r[2], o_r[3]));
    st4(cell.saved + plane + o, make_float4(o_z[0], o_z[1], o_z[2], o_z[3]));
    st4(cell.saved + 2 * plane + o, make_float4(o_n[0], o_n[1], o_n[2], o_n[3]));
    st4(cell.saved + 3 * plane + o, bhn);
    st4(cell.saved + 4 * plane + o, zero4());
  }
}

template <int VARIANT>
static int launch_gru_fwd_wres(const GruFwdBatch& batch, int count, int d, float lambda, const float* decay_wb, size_t plane, hipStream_t st) {
  typedef EpiGruCell<VARIANT> Epi;
  PanelBatch<Epi> pb;
  long long rows = 0;
  for (int i = 0; i < PANEL_MAXP; ++i) pb.p[i] = PanelProblem<Epi>{0, nullptr, nullptr, nullptr, Epi{batch.c[0], d, plane, lambda, decay_wb}};
  for (int i = 0; i < count; ++i) {
    const GruFwdCell& c = batch.c[i];
    pb.p[i] = PanelProblem<Epi>{c.n, c.prev, c.prev_idx, c.w_hh, Epi{c, d, plane, lambda, decay_wb}};
    rows += c.n > 0 ? c.n : 0;
  }
  WresGeom g;
  if (!wres_plan(3 * d, d, d, d, 1, 1 << 20, &g)) return TEMP_E_UNSUPPORTED;
  if ((size_t)96 * g.ldk * 4 > WRES_LDS_BYTES) return TEMP_E_UNSUPPORTED;
  g.tps = 3; g.n_slices = ceil_div(d, 32); g.tail_store = 3; g.gate_stride = d; g.split = 1;
  // two blocks per CU: two waves per SIMD interleave, one wave's MFMAs covering the other's loads / epilogue
  return launch_wres_one<3, Epi>(K_GRU_FWD, pb, count, g, st, 64);
}

// Gate gradients (pointwise).  dgi: [n, 3D] (torch) or [n, D] (type-1); dgh: [n, 3D]; decv: [n].
// d_h = dh_up (nullable) + d_prev_next[next_idx[row]] (nullable; the gradient flowing back from the
// next window position, gathered through the inverse row map, -1 = none).  dhz (= d_h * z) seeds the
// d_prev GEMM epilogue.
struct GruGatesCell {
  int n; const float* saved; const float* dh_up; const float* d_prev_next; const int32_t* next_idx; const float* dt;
  float* dgi; float* dgh; float* decv; float* dhz;
};
struct GruGatesBatch { GruGatesCell c[GRU_MAXP]; };

template <int VARIANT, int LPR>
__global__ void __launch_bounds__(256) k_gru_bwd_gates(GruGatesBatch batch, int D, size_t plane, float lambda,
                                                       const float* __restrict__ decay_wb) {
  const GruGatesCell& cell = batch.c[blockIdx.y];
  const int n = cell.n;
  const float* __restrict__ saved = cell.saved;
  const float* __restrict__ dh_up = cell.dh_up;
  const float* __restrict__ d_prev_next = cell.d_prev_next;
  const int32_t* __restrict__ next_idx = cell.next_idx;
  const float* __restrict__ dt = cell.dt;
  float* __restrict__ dgi = cell.dgi;
  float* __restrict__ dgh = cell.dgh;
  float* __restrict__ decv = cell.decv;
  float* __restrict__ dhz = cell.dhz;
  // one float4 per thread: LPR (power of two >= D/4, at most 64) lanes per row, 256 / LPR rows per block pass;
  // no per-element division (a runtime divisor costs more than the arithmetic of this kernel)
  const int D4 = D >> 2;
  const int lr = threadIdx.x & (LPR - 1), rsub = threadIdx.x / LPR;
  constexpr int RPB = 256 / LPR;
  for (int row = blockIdx.x * RPB + rsub; row < n; row += gridDim.x * RPB) {
    int nx = -1;
    if (next_idx) nx = next_idx[row];
    if (lr == 0) decv[row] = decay_factor(dt[row], lambda, decay_wb);
    for (int c4 = lr; c4 < D4; c4 += LPR) {
      const size_t i = (size_t)row * D + 4 * c4;
      const float4 rg = ld4(saved + i), zg = ld4(saved + plane + i), ng = ld4(saved + 2 * plane + i), hn = ld4(saved + 3 * plane + i),
                   hd = ld4(saved + 4 * plane + i);
      float4 g = dh_up ? ld4(dh_up + i) : zero4();
      if (nx >= 0) g = add4(g, ld4(d_prev_next + (size_t)nx * D + 4 * c4));
      float4 dr_pre, dz_pre, dn_pre, dhn, gz;
#define TEMP_GATE(c)                                          \
      {                                                       \
        const float dn = g.c * (1.f - zg.c);                  \
        const float dz = g.c * (hd.c - ng.c);                 \
        dn_pre.c = dn * (1.f - ng.c * ng.c);                  \
        dr_pre.c = dn_pre.c * hn.c * rg.c * (1.f - rg.c);     \
        dz_pre.c = dz * zg.c * (1.f - zg.c);                  \
        dhn.c = dn_pre.c * rg.c;                              \
        gz.c = g.c * zg.c;                                    \
      }
      TEMP_GATE(x) TEMP_GATE(y) TEMP_GATE(z) TEMP_GATE(w)
#undef TEMP_GATE
      const size_t b3 = (size_t)row * 3 * D + 4 * c4;
      if (VARIANT == TEMP_GRU_TORCH) {
        st4(dgi + b3, dr_pre);
        st4(dgi + b3 + D, dz_pre);
        st4(dgi + b3 + 2 * D, dn_pre);
      } else {
        st4(dgi + i, dn_pre);
      }
      st4(dgh + b3, dr_pre);
      st4(dgh + b3 + D, dz_pre);
      st4(dgh + b3 + 2 * D, dhn);
      st4(dhz + i, gz);
    }
  }
}

// d_prev = (dgh . W_hh + dh * z) * decay[row]; dh*z was left in `io` by the gates kernel
struct EpiGruDprev {
  const float* decv; float* io; int D;
  struct RowCtx { float dec; };
  __device__ __forceinline__ RowCtx row_ctx(int row) const { RowCtx c; c.dec = decv[row]; return c; }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int row, int col) const { return ld4(io + (size_t)row * D + col); }
  __device__ __forceinline__ void fin4(const RowCtx& c, int row, int col, float4 acc, float4 p) const {
    st4(io + (size_t)row * D + col, scale4(add4(acc, p), c.dec));
  }
};
struct EpiStore {
  float* out; int ldo;
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row_ctx(int) const { return RowCtx(); }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int, int) const { return zero4(); }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4) const { st4(out + (size_t)row * ldo + col, acc); }
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

// out[r, :] = x[r, :] * exp(-dt[r] * lambda): the fixed exponential decay of a previous state as a stand-alone row scale
// (linear recurrence of RRGCNLayer / BiRRGCNLayer, models/RRGCN.py:146, models/BiRRGCN.py:126-129).  Self-adjoint: the
// backward is the same kernel on the gradient.  LPR lanes per row (power of two >= d / 4), one float4 per lane.
template <int LPR>
__global__ void __launch_bounds__(256) k_decay_rows(int n, int D, const float* __restrict__ x, const float* __restrict__ dt, float lambda,
                                                    float* __restrict__ out) {
  const int D4 = D >> 2;
  const int lr = threadIdx.x & (LPR - 1), rsub = threadIdx.x / LPR;
  constexpr int RPB = 256 / LPR;
  for (int row = blockIdx.x * RPB + rsub; row < n; row += gridDim.x * RPB) {
    const float dec = expf(-dt[row] * lambda);
    for (int c4 = lr; c4 < D4; c4 += LPR) st4(out + (size_t)row * D + 4 * c4, scale4(ld4(x + (size_t)row * D + 4 * c4), dec));
  }
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

// ---- GRU gradients of a window chain from the ONE gate-gradient matrix g4 = [n][4d] = [dr | dz | dn_i | dn_h] -------------------
static bool grads_g4_plan(int count, const int* ns, int d, WgArgs* a) {
  if (count <= 0 || count > WG_MAXG || !ns || d <= 0 || !bx_enabled()) return false;
  int max_n = 0;
  long long rows = 0;
  for (int i = 0; i < count; ++i) {
    if (ns[i] <= 0) return false;
    max_n = ns[i] > max_n ? ns[i] : max_n;
    rows += ns[i];
  }
  if (rows < BX_MIN_ROWS) return false;                          // (few rows: the per-GRU fp32 kernels win)
  const int nt = wg_col_tiles(d);
  if (nt < 1 || nt > 8) return false;
  return wg_plan(count, d, max_n, a);
}

template <int NT>
static int launch_gru_wgrad_hx(const WgArgs& a, const WgxKeys& keys, hipStream_t st) {
  static const bool granted = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_wgrad_hx<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, wgx_lds_bytes(NT)) == hipSuccess;
  if (!granted) { (void)hipGetLastError(); return TEMP_E_UNSUPPORTED; }
  TEMP_LAUNCH(K_GRU_WGRAD, (k_gru_wgrad_hx<NT>), dim3(8 * a.per_xcd * a.P), dim3(WG_THREADS), wgx_lds_bytes(NT), st, a, keys);
  hx_count();
  return TEMP_OK;
}

template <int NT>
static int launch_gru_wgrad(const WgArgs& a, hipStream_t st) {
  static const bool granted = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_wgrad<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, wg_lds_bytes(NT)) == hipSuccess;
  if (!granted) { (void)hipGetLastError(); return TEMP_E_UNSUPPORTED; }
  TEMP_LAUNCH(K_GRU_WGRAD, (k_gru_wgrad<NT>), dim3(8 * a.per_xcd * a.P), dim3(WG_THREADS), wg_lds_bytes(NT), st, a);
  return TEMP_OK;
}

extern "C" {

static int launch_gru_fwd_batch(const GruFwdBatch& batch, int count, int d, int variant, bool hoisted, float lambda, const float* decay_wb,
                                size_t plane, hipStream_t st) {
  int max_n = 0;
  for (int i = 0; i < count; ++i) if (batch.c[i].n > max_n) max_n = batch.c[i].n;
  if (max_n <= 0) return TEMP_OK;
  const bool stream_only = option(TEMP_OPT_GRU_STREAM) != 0;   // A/B switch
  if (hoisted && d % 8 == 0 && !stream_only) {
    // window-chain cells: weights-resident GEMM with the grouped gate epilogue (EpiGruCell)
    int rc = (variant == TEMP_GRU_TORCH) ? launch_gru_fwd_wres<TEMP_GRU_TORCH>(batch, count, d, lambda, decay_wb, plane, st)
                                         : launch_gru_fwd_wres<TEMP_GRU_TYPE1>(batch, count, d, lambda, decay_wb, plane, st);
    if (rc != TEMP_E_UNSUPPORTED) return rc;
  }
  dim3 grid(ceil_div(max_n, 128), ceil_div(d, 32), count);
#define TEMP_GRU_FWD(V, H) TEMP_LAUNCH(K_GRU_FWD, (k_gru_fwd<V, H>), grid, dim3(256), 0, st, batch, d, lambda, decay_wb, plane)
  if (variant == TEMP_GRU_TORCH) { if (hoisted) TEMP_GRU_FWD(TEMP_GRU_TORCH, true); else TEMP_GRU_FWD(TEMP_GRU_TORCH, false); }
  else { if (hoisted) TEMP_GRU_FWD(TEMP_GRU_TYPE1, true); else TEMP_GRU_FWD(TEMP_GRU_TYPE1, false); }
#undef TEMP_GRU_FWD
  return launch_status();
}

static int launch_gru_fwd(int n, int d, int variant, const float* x, const float* gi, const float* prev, const int32_t* prev_idx,
                          const float* dt, float lambda, const float* decay_wb, const float* w_ih, const float* w_hh,
                          const float* b_ih, const float* b_hh, float* h_out, float* saved, size_t plane, hipStream_t st) {
  GruFwdBatch batch = {};
  batch.c[0] = GruFwdCell{n, x, gi, prev, prev_idx, dt, w_ih, w_hh, b_ih, b_hh, h_out, saved};
  return launch_gru_fwd_batch(batch, 1, d, variant, gi != nullptr, lambda, decay_wb, plane, st);
}

static int launch_gru_gates_batch(const GruGatesBatch& batch, int count, int d, int variant, size_t plane, float lambda,
                                  const float* decay_wb, hipStream_t st) {
  int max_n = 0;
  for (int i = 0; i < count; ++i) if (batch.c[i].n > max_n) max_n = batch.c[i].n;
  if (max_n <= 0) return TEMP_OK;
  const int d4 = d / 4;
  const int lpr = d4 <= 8 ? 8 : (d4 <= 16 ? 16 : (d4 <= 32 ? 32 : 64));
  int gx = ceil_div(max_n, 256 / lpr);
  if (gx > 4096) gx = 4096;
  dim3 grid(gx, count);
#define TEMP_GATES(V, L) TEMP_LAUNCH(K_GRU_BWD_GATES, (k_gru_bwd_gates<V, L>), grid, dim3(256), 0, st, batch, d, plane, lambda, decay_wb)
#define TEMP_GATES_V(V) do { if (lpr == 8) TEMP_GATES(V, 8); else if (lpr == 16) TEMP_GATES(V, 16); else if (lpr == 32) TEMP_GATES(V, 32); else TEMP_GATES(V, 64); } while (0)
  if (variant == TEMP_GRU_TORCH) TEMP_GATES_V(TEMP_GRU_TORCH); else TEMP_GATES_V(TEMP_GRU_TYPE1);
#undef TEMP_GATES_V
#undef TEMP_GATES
  return launch_status();
}

static int launch_gru_gates(int n, int d, int variant, const float* saved, size_t plane, const float* dh_up, const float* d_prev_next,
                            const int32_t* next_idx, const float* dt, float lambda, const float* decay_wb, float* dgi, float* dgh,
                            float* decv, float* dhz, hipStream_t st) {
  GruGatesBatch batch = {};
  batch.c[0] = GruGatesCell{n, saved, dh_up, d_prev_next, next_idx, dt, dgi, dgh, decv, dhz};
  return launch_gru_gates_batch(batch, 1, d, variant, plane, lambda, decay_wb, st);
}

int temp_gru_fwd(int n, int d, int variant, const float* x, const float* prev, const int32_t* prev_idx, const float* dt, float lambda,
                 const float* decay_wb, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* h_out,
                 float* saved, void* stream) {
  if (n < 0 || d <= 0 || !w_ih || !w_hh || !b_ih || !b_hh) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (n > 0 && (!x || !prev || !dt || !h_out || !saved)) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (n == 0) return TEMP_OK;
  return launch_gru_fwd(n, d, variant, x, nullptr, prev, prev_idx, dt, lambda, decay_wb, w_ih, w_hh, b_ih, b_hh, h_out, saved,
                        (size_t)n * d, (hipStream_t)stream);
}

size_t temp_gru_bwd_workspace(int n, int d, int variant) {
  if (n < 0 || d <= 0) return 0;
  return carve_gru(n, d, variant, nullptr).total;
}

static int gru_weight_grads(int n, int d, int variant, const float* x, const float* hdec, const float* dgi, const float* dgh,
                            const float* w_ih, float* d_x, float* d_w_ih, float* d_w_hh, float* d_b_ih, float* d_b_hh, void* tn,
                            size_t tn_bytes, void* cs, size_t cs_bytes, hipStream_t st) {
  const int gi_w = (variant == TEMP_GRU_TORCH) ? 3 * d : d;
  int rc = TEMP_OK;
  if (d_x) {                       // d_x = dgi . W_ih            (W_ih is [gi_w, d] row-major == [K, N])
    rc = launch_gemm_panel(K_GEMM_GRU_DX, n, d, gi_w, dgi, gi_w, nullptr, w_ih, d, 0, EpiStore{d_x, d}, st);
    if (rc) return rc;
  }
  const bool fuse = gemm_tn_can_fuse_bias(d);      // bias gradients ride along as a ones column of the TN GEMM
  rc = gemm_tn(n, gi_w, d, dgi, gi_w, x, d, d_w_ih, d, tn, tn_bytes, st, fuse ? d_b_ih : nullptr);
  if (rc) return rc;
  if (hdec) {
    rc = gemm_tn(n, 3 * d, d, dgh, 3 * d, hdec, d, d_w_hh, d, tn, tn_bytes, st, fuse ? d_b_hh : nullptr);
    if (rc) return rc;
  } else {                         // every row started from the zero state: hdec = 0, so d_W_hh = 0 and only the bias needs dgh
    if (hipMemsetAsync(d_w_hh, 0, (size_t)3 * d * d * sizeof(float), st) != hipSuccess) return TEMP_E_LAUNCH;
    rc = colsum(n, 3 * d, dgh, 3 * d, d_b_hh, cs, cs_bytes, st);
    if (rc) return rc;
  }
  if (fuse) return TEMP_OK;
  rc = colsum(n, gi_w, dgi, gi_w, d_b_ih, cs, cs_bytes, st);
  if (rc) return rc;
  return hdec ? colsum(n, 3 * d, dgh, 3 * d, d_b_hh, cs, cs_bytes, st) : TEMP_OK;
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
  int rc = launch_gru_gates(n, d, variant, saved, nd, d_h_out, nullptr, nullptr, dt, lambda, decay_wb, w.dgi, w.dgh, w.decv, d_prev, st);
  if (rc) return rc;
  rc = launch_gemm_panel(K_GEMM_GRU_DPREV, n, d, 3 * d, w.dgh, 3 * d, nullptr, w_hh, d, 0, EpiGruDprev{w.decv, d_prev, d}, st);
  if (rc) return rc;
  rc = gru_weight_grads(n, d, variant, x, saved + 4 * nd, w.dgi, w.dgh, w_ih, d_x, d_w_ih, d_w_hh, d_b_ih, d_b_hh, w.tn, w.tn_bytes, w.cs,
                        w.cs_bytes, st);
  if (rc) return rc;
  if (decay_wb) {
    TEMP_LAUNCH(K_DECAY_GRAD, k_decay_grad, dim3(1), dim3(256), 0, st, n, d, d_prev, prev, prev_idx, dt, decay_wb, d_decay_wb);
    rc = launch_status();
  }
  return rc;
}

int temp_decay_rows(int n, int d, const float* x, const float* dt, float lambda, float* out, void* stream) {
  if (n < 0 || d <= 0 || (n > 0 && (!x || !dt || !out))) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (n == 0) return TEMP_OK;
  const int d4 = d / 4;
  const int lpr = d4 <= 8 ? 8 : (d4 <= 16 ? 16 : (d4 <= 32 ? 32 : 64));
  int gx = ceil_div(n, 256 / lpr);
  if (gx > 4096) gx = 4096;
  hipStream_t st = (hipStream_t)stream;
#define TEMP_DECAY(L) TEMP_LAUNCH(K_DECAY_GRAD, (k_decay_rows<L>), dim3(gx), dim3(256), 0, st, n, d, x, dt, lambda, out)
  if (lpr == 8) TEMP_DECAY(8); else if (lpr == 16) TEMP_DECAY(16); else if (lpr == 32) TEMP_DECAY(32); else TEMP_DECAY(64);
#undef TEMP_DECAY
  return launch_status();
}

/* ---- window-batched recurrence (see include/temp_amd.h) ---------------------------------------- */
int temp_gru_input_gates(int n, int d, int variant, const float* x, const float* w_ih, const float* b_ih, float* gi, void* stream) {
  if (n < 0 || d <= 0 || !w_ih || !b_ih || (n > 0 && (!x || !gi))) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  const int gi_w = (variant == TEMP_GRU_TORCH) ? 3 * d : d;
  // gi = x . W_ih^T + b_ih       (W_ih stored [gi_w, d] => transposed B)
  return gemm_add_bias_act(K_GEMM_GRU_GI, n, gi_w, d, x, d, nullptr, w_ih, d, 1, nullptr, 0, nullptr, b_ih, TEMP_ACT_NONE, gi, gi_w,
                           (hipStream_t)stream);
}

int temp_gru_input_gates_gather_multi(int count, const int* ns, int d, int variant, const float* const* xs, const int32_t* const* x_idx,
                                      const float* const* w_ihs, const float* const* b_ihs, float* const* gis, void* stream) {
  return temp_gru_input_gates_gather_multi_keys(count, ns, d, variant, xs, x_idx, nullptr, w_ihs, b_ihs, gis, stream);
}

int temp_gru_input_gates_gather_multi_keys(int count, const int* ns, int d, int variant, const float* const* xs, const int32_t* const* x_idx,
                                           const uint32_t* const* x_keys, const float* const* w_ihs, const float* const* b_ihs, float* const* gis,
                                           void* stream) {
  if (count < 0 || d <= 0 || (count > 0 && (!ns || !xs || !w_ihs || !b_ihs || !gis))) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  const int gi_w = (variant == TEMP_GRU_TORCH) ? 3 * d : d;
  for (int i = 0; i < count; ++i)
    if (ns[i] < 0 || !w_ihs[i] || !b_ihs[i] || (ns[i] > 0 && (!xs[i] || !gis[i]))) return TEMP_E_BADARG;
  for (int i0 = 0; i0 < count; i0 += 4) {                    // four problems per launch
    const int c = count - i0 < 4 ? count - i0 : 4;
    const int rc = gemm_bias_multi(K_GEMM_GRU_GI, c, ns + i0, gi_w, d, xs + i0, x_idx ? x_idx + i0 : nullptr, d, w_ihs + i0, d, b_ihs + i0,
                                   gis + i0, gi_w, (hipStream_t)stream, x_keys ? x_keys + i0 : nullptr);
    if (rc) return rc;
  }
  return TEMP_OK;
}

int temp_gru_input_gates_multi(int count, const int* ns, int d, int variant, const float* const* xs, const float* const* w_ihs,
                               const float* const* b_ihs, float* const* gis, void* stream) {
  return temp_gru_input_gates_gather_multi(count, ns, d, variant, xs, nullptr, w_ihs, b_ihs, gis, stream);
}

int temp_gru_cell_fwd(int n, int d, int variant, const float* gi, const float* prev, const int32_t* prev_idx, const float* dt, float lambda,
                      const float* w_hh, const float* b_hh, float* h_out, float* saved, size_t saved_plane, void* stream) {
  if (n < 0 || d <= 0 || !w_hh || !b_hh) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (n > 0 && (!gi || !prev || !dt || !h_out || !saved || saved_plane < (size_t)n * d)) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (n == 0) return TEMP_OK;
  return launch_gru_fwd(n, d, variant, nullptr, gi, prev, prev_idx, dt, lambda, nullptr, nullptr, w_hh, nullptr, b_hh, h_out, saved,
                        saved_plane, (hipStream_t)stream);
}

int temp_gru_cell_bwd(int n, int d, int variant, const float* saved, size_t saved_plane, const float* dh_up, const float* d_prev_next,
                      const int32_t* next_idx, const float* dt, float lambda, const float* w_hh, float* dgi, float* dgh, float* decv,
                      float* d_prev, void* stream) {
  if (n < 0 || d <= 0 || !w_hh) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (n > 0 && (!saved || !dt || !dgi || !dgh || !decv || !d_prev)) return TEMP_E_BADARG;
  if (next_idx && !d_prev_next) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (n == 0) return TEMP_OK;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_gru_gates(n, d, variant, saved, saved_plane, dh_up, d_prev_next, next_idx, dt, lambda, nullptr, dgi, dgh, decv, d_prev, st);
  if (rc) return rc;
  return launch_gemm_panel(K_GEMM_GRU_DPREV, n, d, 3 * d, dgh, 3 * d, nullptr, w_hh, d, 0, EpiGruDprev{decv, d_prev, d}, st);
}

int temp_gru_cell_fwd_multi(int count, const TempGruCellFwd* cells, int d, int variant, float lambda, size_t saved_plane, void* stream) {
  if (count <= 0 || count > GRU_MAXP || !cells || d <= 0) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  GruFwdBatch batch = {};
  GruZeroBatch zb = {};
  int n_gemm = 0, n_zero = 0, max_zero = 0;
  for (int i = 0; i < count; ++i) {
    const TempGruCellFwd& c = cells[i];
    if (c.n < 0 || !c.w_hh || !c.b_hh) return TEMP_E_BADARG;
    if (c.n > 0 && (!c.gi || !c.dt || !c.h_out || !c.saved || saved_plane < (size_t)c.n * d)) return TEMP_E_BADARG;
    if (!c.prev) {                                    // zero previous state: pointwise cell, no GEMM
      if (c.n > 0) { zb.c[n_zero++] = GruZeroCell{c.n, c.gi, c.b_hh, c.h_out, c.saved}; if (c.n > max_zero) max_zero = c.n; }
      continue;
    }
    batch.c[n_gemm++] = GruFwdCell{c.n, nullptr, c.gi, c.prev, c.prev_idx, c.dt, nullptr, c.w_hh, nullptr, c.b_hh, c.h_out, c.saved};
  }
  hipStream_t st = (hipStream_t)stream;
  if (n_zero) {
    int gx = ceil_div((long long)max_zero * (d / 4), 256);
    if (gx > 2048) gx = 2048;
    if (variant == TEMP_GRU_TORCH) TEMP_LAUNCH(K_GRU_FWD, k_gru_fwd_zero<TEMP_GRU_TORCH>, dim3(gx, n_zero), dim3(256), 0, st, zb, d, saved_plane);
    else TEMP_LAUNCH(K_GRU_FWD, k_gru_fwd_zero<TEMP_GRU_TYPE1>, dim3(gx, n_zero), dim3(256), 0, st, zb, d, saved_plane);
    const int rc = launch_status();
    if (rc) return rc;
  }
  if (!n_gemm) return TEMP_OK;
  return launch_gru_fwd_batch(batch, n_gemm, d, variant, true, lambda, nullptr, saved_plane, st);
}

int temp_gru_cell_bwd_multi(int count, const TempGruCellBwd* cells, int d, int variant, float lambda, size_t saved_plane, void* stream) {
  if (count <= 0 || count > GRU_MAXP || !cells || d <= 0) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  GruGatesBatch gb = {};
  PanelBatch<EpiGruDprev> pb;
  for (int i = 0; i < PANEL_MAXP; ++i) pb.p[i] = PanelProblem<EpiGruDprev>{0, nullptr, nullptr, nullptr, EpiGruDprev{nullptr, nullptr, d}};
  for (int i = 0; i < count; ++i) {
    const TempGruCellBwd& c = cells[i];
    if (c.n < 0 || !c.w_hh) return TEMP_E_BADARG;
    if (c.n > 0 && (!c.saved || !c.dt || !c.dgi || !c.dgh || !c.decv || !c.d_prev)) return TEMP_E_BADARG;
    if (c.next_idx && !c.d_prev_next) return TEMP_E_BADARG;
    gb.c[i] = GruGatesCell{c.n, c.saved, c.dh_up, c.d_prev_next, c.next_idx, c.dt, c.dgi, c.dgh, c.decv, c.d_prev};
    pb.p[i] = PanelProblem<EpiGruDprev>{c.no_prev ? 0 : c.n, c.dgh, nullptr, c.w_hh, EpiGruDprev{c.decv, c.d_prev, d}};
  }
  int rc = launch_gru_gates_batch(gb, count, d, variant, saved_plane, lambda, nullptr, st);
  if (rc) return rc;
  return launch_gemm_panel_multi(K_GEMM_GRU_DPREV, pb, count, d, 3 * d, 3 * d, d, 0, st);
}

size_t temp_gru_weight_grads_workspace(int n, int d, int variant) {
  if (n < 0 || d <= 0) return 0;
  const int gi_w = (variant == TEMP_GRU_TORCH) ? 3 * d : d;
  size_t tn = gemm_tn_workspace(n, 3 * d, d);
  if (gemm_tn_workspace(n, gi_w, d) > tn) tn = gemm_tn_workspace(n, gi_w, d);
  return tn + colsum_workspace(n, 3 * d) + 512;
}

int temp_gru_weight_grads(int n, int d, int variant, const float* x, const float* hdec, const float* dgi, const float* dgh,
                          const float* w_ih, float* d_x, float* d_w_ih, float* d_w_hh, float* d_b_ih, float* d_b_hh, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (n < 0 || d <= 0 || !w_ih || !d_w_ih || !d_w_hh || !d_b_ih || !d_b_hh) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH && variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (n > 0 && (!x || !dgi || !dgh)) return TEMP_E_BADARG;          // hdec NULL: all rows started from the zero state
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < temp_gru_weight_grads_workspace(n, d, variant)) return TEMP_E_WORKSPACE;
  const int gi_w = (variant == TEMP_GRU_TORCH) ? 3 * d : d;
  size_t tn = gemm_tn_workspace(n, 3 * d, d);
  if (gemm_tn_workspace(n, gi_w, d) > tn) tn = gemm_tn_workspace(n, gi_w, d);
  char* base = (char*)workspace;
  return gru_weight_grads(n, d, variant, x, hdec, dgi, dgh, w_ih, d_x, d_w_ih, d_w_hh, d_b_ih, d_b_hh, base, tn, base + tn,
                          colsum_workspace(n, 3 * d), (hipStream_t)stream);
}

size_t temp_gru_weight_grads_multi_workspace(int count, const int* ns, int d, int variant) {
  if (count <= 0 || count > 4 || !ns || d <= 0 || variant != TEMP_GRU_TORCH) return 0;
  int max_n = 0;
  for (int i = 0; i < count; ++i) {
    if (ns[i] <= 0) return 0;
    max_n = ns[i] > max_n ? ns[i] : max_n;
  }
  // 0 = "this shape does not take the one-launch path" (same predicate as the launch: the caller allocates nothing and falls back)
  if (d % 4 || !gemm_tn_multi_bias_supported(2 * count, max_n, 3 * d, d, 3 * d, d)) return 0;
  return gemm_tn_multi_bias_workspace(2 * count, max_n, 3 * d, d);
}

int temp_gru_weight_grads_multi(int count, const int* ns, int d, int variant, const float* const* xs, const float* const* hdecs,
                                const float* const* dgis, const float* const* dghs, const float* const* w_ihs, float* const* d_xs,
                                float* d_w, float* d_b, void* workspace, size_t workspace_bytes, void* stream) {
  if (count <= 0 || count > 4 || !ns || !xs || !hdecs || !dgis || !dghs || !w_ihs || !d_xs || !d_w || !d_b) return TEMP_E_BADARG;
  if (variant != TEMP_GRU_TORCH || d % 4) return TEMP_E_UNSUPPORTED;       // (the type-1 cell's dgi is [n, d]: two shape classes)
  int max_n = 0;
  for (int i = 0; i < count; ++i) {
    if (ns[i] <= 0 || !xs[i] || !hdecs[i] || !dgis[i] || !dghs[i] || !w_ihs[i]) return TEMP_E_UNSUPPORTED;
    max_n = ns[i] > max_n ? ns[i] : max_n;
  }
  hipStream_t st = (hipStream_t)stream;
  // the products first (they decide whether this path applies at all), then the d_x panel launch
  int Ms[8];
  const float* As[8];
  const float* Bs[8];
  for (int i = 0; i < count; ++i) {
    Ms[2 * i] = Ms[2 * i + 1] = ns[i];
    As[2 * i] = dgis[i]; Bs[2 * i] = xs[i];                      // d_W_ih = dgi^T x
    As[2 * i + 1] = dghs[i]; Bs[2 * i + 1] = hdecs[i];           // d_W_hh = dgh^T hdec
  }
  int rc = gemm_tn_multi_bias(2 * count, Ms, 3 * d, d, As, 3 * d, Bs, d, d_w, d_b, workspace, workspace_bytes, st);
  if (rc) return rc;
  PanelBatch<EpiStore> batch;
  int nx = 0;
  for (int i = 0; i < count; ++i)
    if (d_xs[i]) batch.p[nx++] = PanelProblem<EpiStore>{ns[i], dgis[i], nullptr, w_ihs[i], EpiStore{d_xs[i], d}};
  for (int i = nx; i < PANEL_MAXP && nx > 0; ++i) { batch.p[i] = batch.p[0]; batch.p[i].M = 0; }
  if (nx > 0) rc = launch_gemm_panel_multi(K_GEMM_GRU_DX, batch, nx, d, 3 * d, 3 * d, d, 0, st);       // d_x = dgi . W_ih
  return rc ? rc : launch_status();
}

size_t temp_gru_grads_g4_workspace(int count, const int* ns, int d) {
  WgArgs a = {};
  if (!grads_g4_plan(count, ns, d, &a)) return 0;                // 0 = "this shape takes the dgi / dgh calls" (nothing to allocate)
  return wg_workspace(a).total;
}

int temp_gru_grads_g4(int count, const int* ns, int d, const float* const* xs, const float* const* hdecs, const float* const* g4s,
                      const float* const* w_ihs, float* const* d_xs, float* d_w, float* d_b, void* workspace, size_t workspace_bytes,
                      void* stream) {
  return temp_gru_grads_g4_keys(count, ns, d, xs, hdecs, g4s, w_ihs, d_xs, d_w, d_b, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int temp_gru_grads_g4_keys(int count, const int* ns, int d, const float* const* xs, const float* const* hdecs, const float* const* g4s,
                           const float* const* w_ihs, float* const* d_xs, float* d_w, float* d_b, const uint32_t* const* g4_row_keys,
                           const uint32_t* const* g4_col_keys, const uint32_t* const* x_col_keys, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (count <= 0 || count > WG_MAXG || !ns || !xs || !hdecs || !g4s || !w_ihs || !d_xs || !d_w || !d_b) return TEMP_E_BADARG;
  WgArgs a = {};
  if (!grads_g4_plan(count, ns, d, &a)) return TEMP_E_UNSUPPORTED;
  for (int i = 0; i < count; ++i) {
    if (!xs[i] || !hdecs[i] || !g4s[i] || !w_ihs[i]) return TEMP_E_UNSUPPORTED;
    a.g[i] = WgGroup{ns[i], g4s[i], xs[i], hdecs[i]};
  }
  const WgWs ws = wg_workspace(a);
  if (!workspace || workspace_bytes < ws.total) return TEMP_E_WORKSPACE;
  char* base = (char*)workspace;
  a.part = (float*)(base + ws.part); a.bpart = (float*)(base + ws.bpart);
  a.part2 = (float*)(base + ws.part2); a.bpart2 = (float*)(base + ws.bpart2);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  bool hx = hx_enabled() && g4_col_keys != nullptr;
  for (int i = 0; i < count && hx; ++i) hx = g4_col_keys[i] != nullptr;
  // d_x = [dr dz dn_i] . W_ih: the first 3d columns of a g4 row (row stride 4d).  It reads g4 and W_ih only -- nothing the weight
  // gradients write -- so it MAY run on the side stream beside them (side_stream.hpp; a parallel branch of a captured graph).
  // Measured (same-box A/B, S-gdelt): 4 us of a 1.97-ms step -- two matrix-pipe kernels at the power cap share the pipe, only the
  // weight pack, the slice reduction and the tails overlap -- while every per-kernel duration of the branch grows by the time it
  // waits for CUs (k_gemm_hxp 124 -> 357 us in the rocprof statistics).  Off by default: TEMP_DEBUG bit 16 turns it on.
  PanelBatch<EpiStore> batch;
  int nx = 0;
  for (int i = 0; i < count; ++i)
    if (d_xs[i]) batch.p[nx++] = PanelProblem<EpiStore>{ns[i], g4s[i], nullptr, w_ihs[i], EpiStore{d_xs[i], d}, (hx && g4_row_keys) ? g4_row_keys[i] : nullptr};
  for (int i = nx; i < PANEL_MAXP && nx > 0; ++i) { batch.p[i] = batch.p[0]; batch.p[i].M = 0; }
  SideScope side(st);
  const bool beside = nx > 0 && (option(TEMP_OPT_DEBUG) & 0x10000) && side_fork(side);
  if (beside) {
    rc = launch_gemm_panel_multi(K_GEMM_GRU_DX, batch, nx, d, 3 * d, 4 * d, d, 0, side.ss->s);
    if (!side_done(side) || rc) return rc ? rc : TEMP_E_LAUNCH;          // (~SideScope joins or drains the branch)
  }
  if (hx) {
    // f16 arithmetic (gru_wgrad_hx.hpp): the column keys of g4 come from the chain backward; those of x are taken here, one pass
    WgxKeys keys = {};
    unsigned* xk = (unsigned*)(base + ws.xkeys);
    for (int i = 0; i < count; ++i) {
      keys.g[i] = g4_col_keys[i];
      if (x_col_keys && x_col_keys[i]) { keys.x[i] = x_col_keys[i]; continue; }     // (the caller's: e.g. from the gather that wrote x)
      keys.x[i] = xk + (size_t)i * d;
      launch_absmax_keys(ns[i], d, xs[i], d, nullptr, xk + (size_t)i * d, xk + (size_t)count * d + (size_t)i * ABSMAX_BLOCKS * d, st);
    }
    switch (wg_col_tiles(d)) {
      case 1: rc = launch_gru_wgrad_hx<1>(a, keys, st); break;
      case 2: rc = launch_gru_wgrad_hx<2>(a, keys, st); break;
      case 3: rc = launch_gru_wgrad_hx<3>(a, keys, st); break;
      case 4: rc = launch_gru_wgrad_hx<4>(a, keys, st); break;
      case 5: rc = launch_gru_wgrad_hx<5>(a, keys, st); break;
      case 6: rc = launch_gru_wgrad_hx<6>(a, keys, st); break;
      case 7: rc = launch_gru_wgrad_hx<7>(a, keys, st); break;
      default: rc = launch_gru_wgrad_hx<8>(a, keys, st); break;
    }
  } else
  switch (wg_col_tiles(d)) {
    case 1: rc = launch_gru_wgrad<1>(a, st); break;
    case 2: rc = launch_gru_wgrad<2>(a, st); break;
    case 3: rc = launch_gru_wgrad<3>(a, st); break;
    case 4: rc = launch_gru_wgrad<4>(a, st); break;
    case 5: rc = launch_gru_wgrad<5>(a, st); break;
    case 6: rc = launch_gru_wgrad<6>(a, st); break;
    case 7: rc = launch_gru_wgrad<7>(a, st); break;
    default: rc = launch_gru_wgrad<8>(a, st); break;
  }
  if (rc) return rc;
  const int Ka = 3 * d, R0 = a.tail ? 256 * a.fb : Ka, S2 = a.tail ? a.S * a.P : 0;
  const long long quads = (long long)2 * count * Ka * (d / 4);
  int gx = ceil_div(quads, 256);
  if (gx > 2048) gx = 2048;
  TEMP_LAUNCH(K_REDUCE_SLICES, k_gru_wgrad_reduce, dim3(gx), dim3(256), 0, st, 2 * count, Ka, d, a.S, a.part, a.bpart, R0, S2, a.part2, a.bpart2, d_w, d_b);
  if (beside) {
    const int jr = side.join();
    return jr ? jr : launch_status();
  }
  if (nx > 0) rc = launch_gemm_panel_multi(K_GEMM_GRU_DX, batch, nx, d, 3 * d, 4 * d, d, 0, st);
  return rc ? rc : launch_status();
}

}  // extern "C"
