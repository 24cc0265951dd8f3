// Dense fp32 pieces of the snapshot encoder on MFMA (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD):
//   * row-panel GEMM  C = epi(A[M,K] . B[K,N])   -- self-loop message (models/RGCN.py:57,80 of the
//     TeMP reference), its d/dh, and the GRU d/dx, d/dh GEMMs, with fused epilogues;
//   * TN split-K GEMM out[Ka,Nb] = sum_m A[m,ka] B[m,nb] -- every weight gradient;
//   * column sums (bias gradients), ReLU mask, row gather / scatter-add, copy probe.
//
// Layout facts used (cdna_hip_programming.md section 3): for mfma_f32_32x32x2f32 lane l supplies
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; result register r of lane l is
// C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
#include "common.hpp"
#include <type_traits>
#include <mutex>
#include <atomic>
#include <cstdlib>
#include "gemm_wres.hpp"
#include "gemm_tn_bx.hpp"

namespace temp {

// ---------------------------------------------------------------------------------------------
// Row-panel GEMM.  A wave owns 32 rows x (NT*32) columns; the 4 waves of a block own 4
// consecutive row tiles and share the B chunk staged in LDS.  A is read straight from global
// memory: the MFMA sums over k in any order, so lane (row i, half hh) loads ONE float4 holding
// k = k0 + 4*hh .. +3 and feeds its 4 components to 4 consecutive MFMAs whose B operand uses the
// same k -- a 16-byte load per lane per 4 MFMAs and no LDS traffic for A.
// ---------------------------------------------------------------------------------------------
struct EpiAddBiasAct {
  const float* addend; int ld_add; const int32_t* row_mask; const float* bias; int act; float* out; int ldo;
  struct RowCtx { int add; };              // (an int: a one-byte struct travelled through scratch memory in gemm_bxr.hpp)
  __device__ __forceinline__ RowCtx row_ctx(int row) const {
    RowCtx c;
    c.add = !addend ? 0 : (row_mask ? row_mask[row] : 1);   // the raw mask word (tested as > 0 where it is used: no wait for the load here)
    return c;
  }
  // pre4(): branch-free float4 loads (row/col clamped by the caller); fin4(): arithmetic + float4 store
  __device__ __forceinline__ float4 pre4(const RowCtx& c, int row, int col) const {
    float4 a = zero4();
    if (addend) {                                   // kernel-uniform
      const float4 v = ld4(addend + (size_t)row * ld_add + col);
      a = c.add > 0 ? v : zero4();
    }
    if (bias) a = add4(a, ld4(bias + col));
    return a;
  }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4 p) const {
    float4 v = add4(acc, p);
    if (act == TEMP_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    st4(out + (size_t)row * ldo + col, v);
  }
  // gemm_bxr.hpp (EpiRawPre): pre4 = (c.add ? raw4 : 0) + bias, taken apart so that a wave can issue all sixteen raw loads of a
  // panel back to back and apply the row mask and the bias (staged in LDS once per block) afterwards
  static constexpr int k_raw_pre = 1;
  __device__ __forceinline__ bool has_addend() const { return addend != nullptr; }
  __device__ __forceinline__ bool has_row_mask() const { return addend && row_mask; }   // else row_ctx() is the same for every row
  __device__ __forceinline__ float4 raw4(int row, int col) const { return ld4(addend + (size_t)row * ld_add + col); }
  __device__ __forceinline__ float bias1(int col) const { return bias ? bias[col] : 0.f; }
};

template <>
struct EpiAccInit<EpiAddBiasAct> { static constexpr bool value = true; };   // fin4 = store(act(acc + pre)): gemm_bxr.hpp starts the accumulators at pre

// Same epilogue with dropout of the product term (the self-loop message in training mode): a separate type, so the
// common no-dropout instantiations carry neither the extra kernel arguments nor the hash code.
struct EpiAddBiasActDrop : EpiAddBiasAct {
  DropSpec drop;
  __device__ __forceinline__ void fin4(const RowCtx& c, int row, int col, float4 acc, float4 p) const {
    EpiAddBiasAct::fin4(c, row, col, drop4(drop, (unsigned)row, (unsigned)col, acc), p);
  }
};

int gemm_add_bias_act(int kid, int M, int N, int K, const float* A, int lda, const int32_t* a_idx, const float* B, int ldb, int trans_b,
                      const float* addend, int ld_add, const int32_t* row_mask, const float* bias, int act, float* out, int ldo,
                      hipStream_t st, const DropSpec* drop) {
  EpiAddBiasAct epi{addend, ld_add, row_mask, bias, act, out, ldo};
  if (drop && drop->p > 0.f) {
    EpiAddBiasActDrop ed;
    static_cast<EpiAddBiasAct&>(ed) = epi;
    ed.drop = *drop;
    return launch_gemm_panel(kid, M, N, K, A, lda, a_idx, B, ldb, trans_b, ed, st);
  }
  return launch_gemm_panel(kid, M, N, K, A, lda, a_idx, B, ldb, trans_b, epi, st);
}

int gemm_bias_multi(int kid, int count, const int* Ms, int N, int K, const float* const* As, const int32_t* const* a_idxs, int lda,
                    const float* const* Bs, int ldb, const float* const* biases, float* const* outs, int ldo, hipStream_t st,
                    const unsigned* const* a_keys) {
  if (count <= 0 || count > PANEL_MAXP) return TEMP_E_BADARG;
  PanelBatch<EpiAddBiasAct> batch;
  for (int i = 0; i < PANEL_MAXP; ++i) {
    const int k = i < count ? i : 0;
    const EpiAddBiasAct epi{nullptr, 0, nullptr, biases[k], TEMP_ACT_NONE, outs[k], ldo};
    batch.p[i] = PanelProblem<EpiAddBiasAct>{i < count ? Ms[k] : 0, As[k], a_idxs ? a_idxs[k] : nullptr, Bs[k], epi, a_keys ? a_keys[k] : nullptr};
  }
  return launch_gemm_panel_multi(kid, batch, count, N, K, lda, ldb, 1, st);
}

__global__ void __launch_bounds__(256) k_mask_rows(size_t n4, int d4, const float4* __restrict__ src, float4* __restrict__ dst, DropSpec drop) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned row = (unsigned)(i / (unsigned)d4), c4 = (unsigned)(i - (size_t)row * d4);
    dst[i] = drop4(drop, row, c4 * 4, src[i]);
  }
}

int mask_rows(int n, int d, const float* src, float* dst, const DropSpec& drop, hipStream_t st) {
  if (n <= 0) return TEMP_OK;
  const size_t n4 = (size_t)n * (d / 4);
  int grid = ceil_div((long long)n4, 256);
  if (grid > 4096) grid = 4096;
  TEMP_LAUNCH(K_RELU_BWD, k_mask_rows, dim3(grid), dim3(256), 0, st, n4, d / 4, (const float4*)src, (float4*)dst, drop);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// TN split-K: out[Ka,Nb] = sum_m A[m,ka] * B[m,nb]  (every weight gradient).
// These GEMMs have a tiny output and two long, thin inputs, so they are bound by how often A and B
// are re-read: a block therefore owns a (WPB*32) x (NT*32) output tile -- for D = 200 ALL 224 padded
// columns and 224/256 rows -- over one slice of m, so A is streamed once and B once per 256 rows of
// Ka.  Wave w owns ka rows [32w, 32w+32) x NT column tiles (NT*16 accumulator registers).  Chunks of
// TN_MC rows of A and B are staged in LDS (coalesced float4 row reads, register-prefetched one chunk
// ahead, double-buffered); for the row pair (m, m+1) lane (li, hh) reads As[m+hh][32w+li] and
// Bs[m+hh][32t+li] -- conflict-free b32 reads.  MFMA operands are swapped so that a lane holds one
// output row and 4 consecutive columns per register quad (float4 stores).  Slices go to a workspace
// and are summed in slice order by k_reduce_slices (deterministic).
// ---------------------------------------------------------------------------------------------
#define TN_MC 16
// SPLIT = 2: the block's WPB waves are KT = WPB/2 row tiles x 2 column halves (wave w and w + KT share a row tile and,
// by the hardware's cyclic wave -> SIMD placement, a SIMD): for NT = 7 a SIMD runs a 4-tile and a 3-tile wave, i.e.
// 7 tiles on every SIMD, where 7 full-width waves load the four SIMDs 2 : 2 : 2 : 1.
template <int NT, int WPB, int SPLIT>
__device__ __forceinline__ void tn_body(int M, int Ka, int Nb, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                        int rows_per_slice, int nb_base, float* __restrict__ part, float* __restrict__ bias_part,
                                        const int slice) {
  constexpr int KT = WPB / SPLIT, NT_LO = (NT + SPLIT - 1) / SPLIT;
  constexpr int T = WPB * 64, BK = KT * 32, BN = NT * 32;
  constexpr int NVA = (TN_MC * BK / 4 + T - 1) / T;          // = 2
  constexpr int NVB = (TN_MC * BN / 4 + T - 1) / T;
  __shared__ __attribute__((aligned(16))) float As[2][TN_MC * BK];
  __shared__ __attribute__((aligned(16))) float Bs[2][TN_MC * BN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int ka_blk = blockIdx.x * BK;
  const int kt = wave % KT, part_id = wave / KT;
  const int t_beg = part_id * NT_LO;                         // first column tile of this wave
  const int ka0 = ka_blk + kt * 32;
  const int nb0 = nb_base + blockIdx.y * BN;
  const int mbeg = slice * rows_per_slice, mend = min(M, mbeg + rows_per_slice);
  const bool wave_on = ka0 < Ka;
  f32x16 acc[NT_LO];
#pragma unroll
  for (int t = 0; t < NT_LO; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // fetch: raw float4 loads (out-of-range pieces read element 0, a valid address); the zero-fill
  // select happens in store(), AFTER the MFMAs of the current chunk, so the loads of chunk c+1 stay
  // in flight behind them instead of being waited for at issue.
  float4 ra[NVA], rb[NVB];
  auto a_ok = [&](int i, int m0) {
    const int p = threadIdx.x + i * T;
    const int r = p / (BK / 4), c = (p - r * (BK / 4)) << 2;
    return (p < TN_MC * BK / 4) && (m0 + r < mend) && (ka_blk + c < Ka);
  };
  auto b_inr = [&](int i, int m0) {
    const int p = threadIdx.x + i * T;
    const int r = p / (BN / 4);
    return (p < TN_MC * BN / 4) && (m0 + r < mend);
  };
  auto fetch = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int p = threadIdx.x + i * T;
      const int r = p / (BK / 4), c = (p - r * (BK / 4)) << 2;
      ra[i] = ld4(A + (a_ok(i, m0) ? (unsigned)((m0 + r) * lda + ka_blk + c) : 0u));
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int p = threadIdx.x + i * T;
      const int r = p / (BN / 4), c = (p - r * (BN / 4)) << 2;
      const bool ok = b_inr(i, m0) && (nb0 + c < Nb);
      rb[i] = ld4(B + (ok ? (unsigned)((m0 + r) * ldb + nb0 + c) : 0u));
    }
  };
  auto store = [&](int buf, int m0) {
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int p = threadIdx.x + i * T;
      if (p < TN_MC * BK / 4) st4(&As[buf][p << 2], a_ok(i, m0) ? ra[i] : zero4());
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int p = threadIdx.x + i * T;
      const int r = p / (BN / 4), c = (p - r * (BN / 4)) << 2;
      const bool inr = b_inr(i, m0);
      float4 v = (inr && (nb0 + c < Nb)) ? rb[i] : zero4();
      // bias gradient for free: the first padding column of B is a column of ones, so output column
      // Nb is sum_m A[m, ka]
      if (bias_part && inr && (nb0 + c == Nb)) v.x = 1.f;
      if (p < TN_MC * BN / 4) st4(&Bs[buf][p << 2], v);
    }
  };

  const int nchunks = (mend > mbeg) ? (mend - mbeg + TN_MC - 1) / TN_MC : 0;
  if (nchunks > 0) {
    fetch(mbeg);
    store(0, mbeg);
  }
  __syncthreads();
  // the whole main loop + epilogue, instantiated per tile count of the wave (NTW is a compile-time constant, so the
  // MFMA body has no guards; both instantiations execute the same sequence of barriers)
  auto run = [&](auto ntw_c) {
    constexpr int NTW = decltype(ntw_c)::value;
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 1 < nchunks;
      fetch(mbeg + (c + 1) * TN_MC);           // unconditional (past the end: clamped re-reads) so no value merge forces a wait here
      __builtin_amdgcn_sched_barrier(0);
      if (wave_on) {
        const float* as = As[c & 1] + kt * 32 + li + hh * BK;
        const float* bs = Bs[c & 1] + t_beg * 32 + li + hh * BN;
        // LDS operands of row pair mp+2 are read while the MFMAs of row pair mp issue
        float a_cur = as[0], b_cur[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) b_cur[t] = bs[t * 32];
#pragma unroll
        for (int mp = 0; mp < TN_MC; mp += 2) {
          float a_nxt = 0.f, b_nxt[NTW];
          if (mp + 2 < TN_MC) {
            a_nxt = as[(mp + 2) * BK];
#pragma unroll
            for (int t = 0; t < NTW; ++t) b_nxt[t] = bs[(mp + 2) * BN + t * 32];
          }
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur[t], a_cur, acc[t], 0, 0, 0);
          if (mp + 2 < TN_MC) {
            a_cur = a_nxt;
#pragma unroll
            for (int t = 0; t < NTW; ++t) b_cur[t] = b_nxt[t];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) store((c + 1) & 1, mbeg + (c + 1) * TN_MC);
      __syncthreads();
    }
    // swapped operands: lane (li, hh) holds output row ka0 + li, register quad q of tile t holds the
    // four columns nb0 + (t_beg + t)*32 + 8q + 4hh .. +3
    const int row = ka0 + li;
    if (!wave_on || row >= Ka) return;
    float* p = part + (size_t)slice * Ka * Nb + (size_t)row * Nb;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = nb0 + (t_beg + t) * 32 + 8 * q + 4 * hh;
        if (col < Nb) st4(p + col, make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]));
        else if (bias_part && col == Nb) bias_part[(size_t)slice * Ka + row] = acc[t][4 * q];
      }
    }
  };
  if constexpr (SPLIT == 1) {
    run(std::integral_constant<int, NT>());
  } else {
    if (part_id == 0) run(std::integral_constant<int, NT_LO>());
    else run(std::integral_constant<int, NT - NT_LO>());
  }
}

template <int NT, int WPB, int SPLIT = 1>
__global__ void __launch_bounds__(WPB * 64) k_gemm_tn(int M, int Ka, int Nb, const float* __restrict__ A, int lda,
                                                      const float* __restrict__ B, int ldb, int rows_per_slice, int nb_base,
                                                      float* __restrict__ part, float* __restrict__ bias_part) {
  tn_body<NT, WPB, SPLIT>(M, Ka, Nb, A, lda, B, ldb, rows_per_slice, nb_base, part, bias_part, blockIdx.z);
}

// Several products of one shape class in ONE launch (the per-window d_all = d_scores^T . q of the loss: eight launches of a few
// dozen row blocks each were latency chains on an idle chip): blockIdx.z = problem * slices + slice.
#define TN_MAXP 8
struct TnBatch { int M[TN_MAXP]; const float* A[TN_MAXP]; const float* B[TN_MAXP]; float* part[TN_MAXP]; };
template <int NT, int WPB, int SPLIT = 1>
__global__ void __launch_bounds__(WPB * 64) k_gemm_tn_multi(TnBatch b, int Ka, int Nb, int lda, int ldb, int rows_per_slice, int slices) {
  const int prob = blockIdx.z / slices, slice = blockIdx.z - prob * slices;
  tn_body<NT, WPB, SPLIT>(b.M[prob], Ka, Nb, b.A[prob], lda, b.B[prob], ldb, rows_per_slice, 0, b.part[prob], nullptr, slice);
}

// out = sum over n_slices of part[s] (slice stride `elems`), optionally a SECOND, flat array in the same launch (the bias partials
// of a weight-gradient product: one launch per product instead of two)
__global__ void __launch_bounds__(256) k_reduce_slices(int n_slices, size_t elems, int width, const float* __restrict__ part,
                                                       float* __restrict__ out, int ldo, size_t elems2, const float* __restrict__ part2,
                                                       float* __restrict__ out2) {
  // elems % 4 == 0 is guaranteed by the callers (all widths are multiples of 4): float4 lanes,
  // 16 slice loads in flight (a thread's loop is a chain of memory round trips: 48 slices = 3 of them), summed in slice order
  // (deterministic)
  const size_t e4 = elems >> 2, e4b = elems2 >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < e4 + e4b; i += (size_t)gridDim.x * blockDim.x) {
    const bool second = i >= e4;
    const size_t stride = second ? elems2 : elems;
    const size_t ii = second ? i - e4 : i;
    const float* p = (second ? part2 : part) + (ii << 2);
    float4 acc = zero4();
    int s = 0;
    for (; s + 16 <= n_slices; s += 16) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = ld4(p + (size_t)(s + u) * stride);
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = add4(acc, v[u]);
    }
    for (; s + 4 <= n_slices; s += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld4(p + (size_t)(s + u) * stride);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = add4(acc, v[u]);
    }
    for (; s < n_slices; ++s) acc = add4(acc, ld4(p + (size_t)s * stride));
    const size_t e = ii << 2;
    if (second) { st4(out2 + e, acc); continue; }
    const size_t r = e / width, c = e - r * width;
    st4(out + r * ldo + c, acc);
  }
}

void reduce_slices(int n_slices, size_t elems, int width, const float* part, float* out, int ldo, hipStream_t st, size_t elems2, const float* part2,
                   float* out2) {
  int rg = ceil_div((long long)(elems + elems2) / 4, 256);
  if (rg > 2048) rg = 2048;
  if (rg < 1) rg = 1;
  TEMP_LAUNCH(K_REDUCE_SLICES, k_reduce_slices, dim3(rg), dim3(256), 0, st, n_slices, elems, width, part, out, ldo, elems2, part2, out2);
}

struct TnCfg { int wpb, bk, kab, nt, nbb, slices, split; };
static TnCfg tn_cfg(int M, int Ka, int Nb) {
  TnCfg c;
  const int ktiles = ceil_div(Ka, 32);
  // 7- or 8-wave blocks (one per CU), whichever pads Ka less: D = 200 -> 7 row tiles, one block owns all
  // of Ka; 3D = 600 -> 19 tiles = 7 + 7 + 5.  (4-wave blocks, two per CU, measured 20 % slower: B is
  // staged twice as often.)
  c.wpb = (ceil_div(ktiles, 7) * 7 - ktiles <= ceil_div(ktiles, 8) * 8 - ktiles) ? 7 : 8;
  c.split = 1;
  const int ntiles_n = ceil_div(Nb, 32);
  const int split_off = !option(TEMP_OPT_TN_SPLIT);
  if (ntiles_n >= 5 && ntiles_n <= 7 && !split_off) { c.wpb = 8; c.split = 2; }      // 4 row tiles x (4 + 3) column tiles per block
  c.bk = (c.wpb / c.split) * 32;
  c.kab = ceil_div(Ka, c.bk);
  const int ntiles = ceil_div(Nb, 32);
  c.nt = ntiles >= 5 ? 7 : (ntiles >= 3 ? 4 : (ntiles == 2 ? 2 : 1));
  c.nbb = ceil_div(ntiles, c.nt);
  long long s = 256 / ((long long)c.kab * c.nbb);     // ~ one 7/8-wave block per CU
  const long long max_s = M < 2048 ? (M + 31) / 32 : (M + 127) / 128;   // at least 128 rows per slice (32 for a tiny M: the launch is all latency)
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  c.slices = (int)s;
  return c;
}

size_t gemm_tn_workspace(int M, int Ka, int Nb) {
  const TnCfg c0 = tn_cfg(M, Ka, Nb);
  size_t S = (size_t)c0.slices;
  if (c0.split == 2 && c0.nbb == 1) {                          // any of the kernels fits
    size_t sb = (size_t)tn_bx_slices(M, c0.kab);
    if (sb > S) S = sb;
    sb = (size_t)tn_bx8_slices(M, ceil_div(Ka, 256));
    if (sb > S) S = sb;
  }
  // (a product over more rows than 32-bit element offsets reach runs in row chunks, each with its own slices: tn_bx_chunks; 200-
  //  wide operands reach that at 10.7 M rows -- leading dimensions up to 1024 floats are provided for)
  const size_t chunks = (size_t)((long long)M * 1024 >= (1ll << 31) ? ceil_div((long long)M * 1024, (1ll << 31) - 1) : 1);
  S *= chunks;
  return align_up(S * Ka * Nb * sizeof(float), 256) + align_up(S * Ka * sizeof(float), 256);
}

// bias_out (nullable): also produce bias_out[ka] = sum_m A[m, ka] (needs Nb % 32 != 0: a padding column exists)
bool gemm_tn_can_fuse_bias(int Nb) { return (Nb % 32) != 0; }

template <int NT>
static void launch_tn(const TnCfg& c, int M, int Ka, int Nb, const float* A, int lda, const float* B, int ldb, int rps, float* part,
                      float* bpart, hipStream_t st) {
  dim3 grid(c.kab, c.nbb, c.slices);
  if (c.split == 2) {
    if constexpr (NT == 7) TEMP_LAUNCH(K_GEMM_TN, (k_gemm_tn<7, 8, 2>), grid, dim3(8 * 64), 0, st, M, Ka, Nb, A, lda, B, ldb, rps, 0, part, bpart);
    return;
  }
  if (c.wpb == 7) TEMP_LAUNCH(K_GEMM_TN, (k_gemm_tn<NT, 7>), grid, dim3(7 * 64), 0, st, M, Ka, Nb, A, lda, B, ldb, rps, 0, part, bpart);
  else TEMP_LAUNCH(K_GEMM_TN, (k_gemm_tn<NT, 8>), grid, dim3(8 * 64), 0, st, M, Ka, Nb, A, lda, B, ldb, rps, 0, part, bpart);
}

int gemm_tn(int M, int Ka, int Nb, const float* A, int lda, const float* B, int ldb, float* out, int ldo, void* ws, size_t ws_bytes,
            hipStream_t st, float* bias_out) {
  if (Ka <= 0 || Nb <= 0) return TEMP_OK;
  if (Ka % 4 || Nb % 4 || lda % 4 || ldb % 4) return TEMP_E_UNSUPPORTED;
  if (bias_out && !gemm_tn_can_fuse_bias(Nb)) return TEMP_E_UNSUPPORTED;
  const TnCfg c = tn_cfg(M, Ka, Nb);
  const bool use_bx = c.split == 2 && c.nbb == 1 && tn_bx_ok(M, Ka, Nb, lda, ldb);
  const bool bx8 = use_bx && tn_bx8_ok(Ka);
  const int kab_bx = bx8 ? ceil_div(Ka, 256) : c.kab;
  const int S = use_bx ? (bx8 ? tn_bx8_slices(M, kab_bx) : tn_bx_slices(M, c.kab)) : c.slices;
  if (ws_bytes < gemm_tn_workspace(M, Ka, Nb) || !ws) return TEMP_E_WORKSPACE;
  int rps = ceil_div(M > 0 ? M : 1, S);
  rps = (rps + TN_MC - 1) / TN_MC * TN_MC;
  float* part = (float*)ws;
  float* bpart = bias_out ? (float*)((char*)ws + align_up((size_t)S * Ka * Nb * sizeof(float), 256)) : nullptr;
  const int chunks = use_bx ? tn_bx_chunks(M, Ka, lda, ldb) : 1;
  if (use_bx && chunks > 1) {
    // row chunks, each a launch of its own over S slices of ITS rows; the reduction below then sums chunks x S slices in order
    if ((lda > 1024 || ldb > 1024)) return TEMP_E_UNSUPPORTED;       // (the workspace query provides for leading dimensions <= 1024)
    const int rows_c = ceil_div(ceil_div(M, chunks), 256) * 256;
    int done = 0, total_s = 0;
    for (int ci = 0; ci < chunks && done < M; ++ci) {
      const int mc = M - done < rows_c ? M - done : rows_c;
      const int Sc = bx8 ? tn_bx8_slices(mc, kab_bx) : tn_bx_slices(mc, c.kab);
      int rc_ = ceil_div(mc, Sc);
      rc_ = (rc_ + TN_MC - 1) / TN_MC * TN_MC;
      float* bp = bias_out ? (float*)((char*)ws + align_up((size_t)chunks * S * Ka * Nb * sizeof(float), 256)) + (size_t)total_s * Ka : nullptr;
      launch_tn_bx(mc, Ka, Nb, A + (size_t)done * lda, lda, B + (size_t)done * ldb, ldb, rc_, kab_bx, Sc, part + (size_t)total_s * Ka * Nb, bp, st);
      done += mc;
      total_s += Sc;
    }
    float* bp0 = bias_out ? (float*)((char*)ws + align_up((size_t)chunks * S * Ka * Nb * sizeof(float), 256)) : nullptr;
    reduce_slices(total_s, (size_t)Ka * Nb, Nb, part, out, ldo, st, bias_out ? (size_t)Ka : 0, bp0, bias_out);
    return launch_status();
  }
  if (use_bx) launch_tn_bx(M, Ka, Nb, A, lda, B, ldb, rps, kab_bx, S, part, bpart, st);
  else if (c.nt == 7) launch_tn<7>(c, M, Ka, Nb, A, lda, B, ldb, rps, part, bpart, st);
  else if (c.nt == 4) launch_tn<4>(c, M, Ka, Nb, A, lda, B, ldb, rps, part, bpart, st);
  else if (c.nt == 2) launch_tn<2>(c, M, Ka, Nb, A, lda, B, ldb, rps, part, bpart, st);
  else launch_tn<1>(c, M, Ka, Nb, A, lda, B, ldb, rps, part, bpart, st);
  reduce_slices(S, (size_t)Ka * Nb, Nb, part, out, ldo, st, bias_out ? (size_t)Ka : 0, bpart, bias_out);      // weights + bias partials in one launch
  return launch_status();
}

// count <= TN_MAXP products out_i[Ka,Nb] = A_i[M_i,Ka]^T . B_i[M_i,Nb] (same Ka, Nb, leading dimensions) in one launch of the
// fp32 kernel; with one slice per problem the blocks write out_i directly (ldo == Nb), else partials + one reduction per problem.
// Returns TEMP_E_UNSUPPORTED for shapes it does not take (the caller then loops over gemm_tn).
// slices per problem of the split-operand multi launch: enough blocks for the chip over all problems, a multiple of 8 (XCD mapping)
static int tn_bx8_multi_slices(int count, int max_m, int kab8) {
  int S = ceil_div(256, (count > 0 ? count : 1) * kab8);
  S = (S + 7) / 8 * 8;
  const int cap = tn_bx8_slices(max_m, kab8);
  if (S > cap) S = cap;
  return S < 1 ? 1 : S;
}

size_t gemm_tn_multi_workspace(int count, int max_m, int Ka, int Nb) {
  const TnCfg c = tn_cfg(max_m, Ka, Nb);
  long long s = 256 / ((long long)c.kab * c.nbb * (count > 0 ? count : 1));
  if (s > c.slices) s = c.slices;
  if (s < 1) s = 1;
  size_t w = s > 1 ? align_up((size_t)count * s * Ka * Nb * sizeof(float), 256) : 0;
  if (tn_bx8_ok(Ka)) {                                          // the split-operand kernel's launch (large M)
    const size_t wb = align_up((size_t)count * tn_bx8_multi_slices(count, max_m, ceil_div(Ka, 256)) * Ka * Nb * sizeof(float), 256);
    if (wb > w) w = wb;
  }
  return w;
}
int gemm_tn_multi(int count, const int* Ms, int Ka, int Nb, const float* const* As, int lda, const float* const* Bs, int ldb, float* const* outs,
                  int ldo, void* ws, size_t ws_bytes, hipStream_t st) {
  if (count <= 0 || count > TN_MAXP || Ka <= 0 || Nb <= 0) return TEMP_E_UNSUPPORTED;
  if (Ka % 4 || Nb % 4 || lda % 4 || ldb % 4 || ldo != Nb) return TEMP_E_UNSUPPORTED;
  int max_m = 0;
  for (int i = 0; i < count; ++i) max_m = Ms[i] > max_m ? Ms[i] : max_m;
  if (max_m <= 0) return TEMP_E_UNSUPPORTED;
  const TnCfg c = tn_cfg(max_m, Ka, Nb);
  if (c.split == 2 && c.nbb == 1 && tn_bx_ok(max_m, Ka, Nb, lda, ldb)) {                             // large M: the split-operand kernel
    if (!tn_bx8_ok(Ka)) return TEMP_E_UNSUPPORTED;
    const int kab8 = ceil_div(Ka, 256), S = tn_bx8_multi_slices(count, max_m, kab8);
    if (!ws || ws_bytes < gemm_tn_multi_workspace(count, max_m, Ka, Nb)) return TEMP_E_WORKSPACE;
    int rps = ceil_div(max_m, S);
    rps = (rps + TN_MC - 1) / TN_MC * TN_MC;
    // outputs that follow each other in memory (the windows' blocks of one gradient matrix): the partials of a slice lie side by
    // side ([slice][product][Ka * Nb]) and ONE reduction sums them (count launches of a few microseconds otherwise)
    bool adjacent = count > 1;
    for (int i = 1; i < count; ++i) adjacent = adjacent && outs[i] == outs[i - 1] + (size_t)Ka * Nb;
    TnBxBatch b;
    for (int i = 0; i < 8; ++i) {
      const int k = i < count ? i : 0;
      b.M[i] = i < count ? Ms[k] : 0; b.A[i] = As[k]; b.B[i] = Bs[k];
      b.part[i] = (float*)ws + (adjacent ? (size_t)k * Ka * Nb : (size_t)k * S * Ka * Nb);
      b.bpart[i] = nullptr;
    }
    b.pstride = (adjacent ? (size_t)count : (size_t)1) * Ka * Nb; b.bstride = (size_t)Ka;
    const int bpp = 8 * ceil_div(S, 8) * kab8, nt = ceil_div(Nb, 32);
    if (nt == 7) TEMP_LAUNCH(K_GEMM_TN, (k_gemm_tn_bx8_multi<7>), dim3(bpp * count), dim3(TNBX_THREADS), 0, st, b, Ka, Nb, lda, ldb, rps, kab8, S, bpp);
    else if (nt == 6) TEMP_LAUNCH(K_GEMM_TN, (k_gemm_tn_bx8_multi<6>), dim3(bpp * count), dim3(TNBX_THREADS), 0, st, b, Ka, Nb, lda, ldb, rps, kab8, S, bpp);
    else TEMP_LAUNCH(K_GEMM_TN, (k_gemm_tn_bx8_multi<5>), dim3(bpp * count), dim3(TNBX_THREADS), 0, st, b, Ka, Nb, lda, ldb, rps, kab8, S, bpp);
    if (adjacent) reduce_slices(S, (size_t)count * Ka * Nb, Nb, (float*)ws, outs[0], ldo, st);
    else
      for (int i = 0; i < count; ++i) reduce_slices(S, (size_t)Ka * Nb, Nb, (float*)ws + (size_t)i * S * Ka * Nb, outs[i], ldo, st);
    return launch_status();
  }
  if (!(c.split == 2 && c.nt == 7) && !(c.split == 1 && (c.nt == 7 || c.nt == 4 || c.nt == 2 || c.nt == 1))) return TEMP_E_UNSUPPORTED;
  if ((long long)max_m * (lda > ldb ? lda : ldb) >= (1ll << 31)) return TEMP_E_UNSUPPORTED;
  long long s = 256 / ((long long)c.kab * c.nbb * count);
  if (s > c.slices) s = c.slices;
  if (s < 1) s = 1;
  const int S = (int)s;
  if (S > 1 && (!ws || ws_bytes < gemm_tn_multi_workspace(count, max_m, Ka, Nb))) return TEMP_E_WORKSPACE;
  int rps = ceil_div(max_m, S);
  rps = (rps + TN_MC - 1) / TN_MC * TN_MC;
  TnBatch b;
  for (int i = 0; i < TN_MAXP; ++i) {
    const int k = i < count ? i : 0;
    b.M[i] = i < count ? Ms[k] : 0; b.A[i] = As[k]; b.B[i] = Bs[k];
    b.part[i] = S > 1 ? (float*)ws + (size_t)k * S * Ka * Nb : outs[k];
  }
  dim3 grid(c.kab, c.nbb, S * count);
#define TEMP_TN_MULTI(NT_, W_, SP_) TEMP_LAUNCH(K_GEMM_TN, (k_gemm_tn_multi<NT_, W_, SP_>), grid, dim3(W_ * 64), 0, st, b, Ka, Nb, lda, ldb, rps, S)
  if (c.split == 2) TEMP_TN_MULTI(7, 8, 2);
  else if (c.wpb == 7) { if (c.nt == 7) TEMP_TN_MULTI(7, 7, 1); else if (c.nt == 4) TEMP_TN_MULTI(4, 7, 1); else if (c.nt == 2) TEMP_TN_MULTI(2, 7, 1); else TEMP_TN_MULTI(1, 7, 1); }
  else { if (c.nt == 7) TEMP_TN_MULTI(7, 8, 1); else if (c.nt == 4) TEMP_TN_MULTI(4, 8, 1); else if (c.nt == 2) TEMP_TN_MULTI(2, 8, 1); else TEMP_TN_MULTI(1, 8, 1); }
#undef TEMP_TN_MULTI
  if (S > 1)
    for (int i = 0; i < count; ++i) reduce_slices(S, (size_t)Ka * Nb, Nb, (float*)ws + (size_t)i * S * Ka * Nb, outs[i], ldo, st);
  return launch_status();
}

// count <= 8 weight-gradient products of ONE shape class with their bias column sums (out_i[Ka,Nb] = A_i^T B_i, bias_i[ka] =
// sum_m A_i[m, ka]) in ONE launch of the split-operand kernel and ONE reduction: outs / biases are contiguous ([count][Ka][Nb],
// [count][Ka]) and the partials of a slice lie side by side ([slice][product][Ka*Nb]), so the reduction sees one long array.
// The slices are cut so that all products' blocks fit the chip at once (256 / (count * row blocks) slices per product): the same
// MFMA work as `count` launches, without their boundaries, with count x fewer partials.  TEMP_E_UNSUPPORTED for shapes the
// split-operand kernel does not take (the caller loops over gemm_tn).
static int tn_multi_bias_slices(int count, int max_m, int kab8) {
  // 256 blocks, one per CU: tn_multi_units() (product, slice) pairs of kab8 row blocks each (gemm_tn_bx.hpp)
  int S = tn_multi_units(kab8) / (count > 0 ? count : 1);
  const int cap = tn_bx8_slices(max_m, kab8);
  if (S > cap) S = cap;
  return S < 1 ? 1 : S;
}
// the ONE support predicate of the multi-product launch (workspace query and launch agree: a non-zero size means "supported")
bool gemm_tn_multi_bias_supported(int count, int max_m, int Ka, int Nb, int lda, int ldb) {
  if (count <= 0 || count > 8 || max_m <= 0 || Ka <= 0 || Nb <= 0 || Ka % 4 || Nb % 4 || lda % 4 || ldb % 4) return false;
  const TnCfg c = tn_cfg(max_m, Ka, Nb);
  if (!(c.split == 2 && c.nbb == 1 && tn_bx_ok(max_m, Ka, Nb, lda, ldb) && tn_bx8_ok(Ka) && gemm_tn_can_fuse_bias(Nb))) return false;
  const int nt = ceil_div(Nb, 32);
  return nt >= 5 && nt <= 7 && tn_multi_units(ceil_div(Ka, 256)) >= count;
}
size_t gemm_tn_multi_bias_workspace(int count, int max_m, int Ka, int Nb) {
  const size_t S = (size_t)tn_multi_bias_slices(count, max_m, ceil_div(Ka, 256));
  return align_up(S * count * Ka * Nb * sizeof(float), 256) + align_up(S * count * Ka * sizeof(float), 256);
}
int gemm_tn_multi_bias(int count, const int* Ms, int Ka, int Nb, const float* const* As, int lda, const float* const* Bs, int ldb, float* outs,
                       float* biases, void* ws, size_t ws_bytes, hipStream_t st) {
  int max_m = 0;
  for (int i = 0; i < count && i < 8; ++i) max_m = Ms[i] > max_m ? Ms[i] : max_m;
  if (!gemm_tn_multi_bias_supported(count, max_m, Ka, Nb, lda, ldb)) return TEMP_E_UNSUPPORTED;
  const int nt = ceil_div(Nb, 32);
  const int kab8 = ceil_div(Ka, 256), S = tn_multi_bias_slices(count, max_m, kab8);
  if (!ws || ws_bytes < gemm_tn_multi_bias_workspace(count, max_m, Ka, Nb)) return TEMP_E_WORKSPACE;
  int rps = ceil_div(max_m, S);
  rps = (rps + TN_MC - 1) / TN_MC * TN_MC;
  float* part = (float*)ws;
  float* bpart = (float*)((char*)ws + align_up((size_t)S * count * Ka * Nb * sizeof(float), 256));
  TnBxBatch b;
  for (int i = 0; i < 8; ++i) {
    const int k = i < count ? i : 0;
    b.M[i] = i < count ? Ms[k] : 0; b.A[i] = As[k]; b.B[i] = Bs[k];
    b.part[i] = part + (size_t)k * Ka * Nb; b.bpart[i] = bpart + (size_t)k * Ka;
  }
  b.pstride = (size_t)count * Ka * Nb; b.bstride = (size_t)count * Ka;
  const int grid = 256;                                        // blocks_per_problem = 0: pairs dealt over the XCDs (gemm_tn_bx.hpp)
  if (nt == 7) TEMP_LAUNCH(K_GEMM_TN_BX8, (k_gemm_tn_bx8_multi<7>), dim3(grid), dim3(TNBX_THREADS), 0, st, b, Ka, Nb, lda, ldb, rps, kab8, S, 0);
  else if (nt == 6) TEMP_LAUNCH(K_GEMM_TN_BX8, (k_gemm_tn_bx8_multi<6>), dim3(grid), dim3(TNBX_THREADS), 0, st, b, Ka, Nb, lda, ldb, rps, kab8, S, 0);
  else TEMP_LAUNCH(K_GEMM_TN_BX8, (k_gemm_tn_bx8_multi<5>), dim3(grid), dim3(TNBX_THREADS), 0, st, b, Ka, Nb, lda, ldb, rps, kab8, S, 0);
  reduce_slices(S, (size_t)count * Ka * Nb, Nb, part, outs, Nb, st, (size_t)count * Ka, bpart, biases);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// Column sums: block (cx, cy) sums rows [cy*RPB, ...) of a 64-column strip into a partial, then
// k_reduce_slices adds the partials in order.
// ---------------------------------------------------------------------------------------------
#define CS_RPB 512
__global__ void __launch_bounds__(256) k_colsum_part(int rows, int cols, const float* __restrict__ X, int ldx, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int r0 = blockIdx.y * CS_RPB, r1 = min(rows, r0 + CS_RPB);
  float acc = 0.f;
  if (c < cols)
    for (int r = r0 + w; r < r1; r += 32) {                    // eight row loads in flight per lane (one at a time was all latency)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r + 4 * u < r1 ? X[(size_t)(r + 4 * u) * ldx + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
  red[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && c < cols) part[(size_t)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

size_t colsum_workspace(int rows, int cols) { return align_up((size_t)ceil_div(rows > 0 ? rows : 1, CS_RPB) * cols * sizeof(float), 256); }

int colsum(int rows, int cols, const float* X, int ldx, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
  if (cols <= 0) return TEMP_OK;
  if (cols % 4) return TEMP_E_UNSUPPORTED;
  const int nb = ceil_div(rows > 0 ? rows : 1, CS_RPB);
  if (!ws || ws_bytes < (size_t)nb * cols * sizeof(float)) return TEMP_E_WORKSPACE;
  TEMP_LAUNCH(K_COLSUM, k_colsum_part, dim3(ceil_div(cols, 64), nb), dim3(256), 0, st, rows, cols, X, ldx, (float*)ws);
  TEMP_LAUNCH(K_REDUCE_SLICES, k_reduce_slices, dim3(ceil_div(cols / 4, 256)), dim3(256), 0, st, nb, (size_t)cols, cols, (const float*)ws, out, cols, (size_t)0,
              (const float*)nullptr, (float*)nullptr);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_relu_bwd(size_t n4, const float4* __restrict__ y, const float4* __restrict__ dy,
                                                  float4* __restrict__ dz) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = y[i], g = dy[i];
    dz[i] = make_float4(a.x > 0.f ? g.x : 0.f, a.y > 0.f ? g.y : 0.f, a.z > 0.f ? g.z : 0.f, a.w > 0.f ? g.w : 0.f);
  }
}
int relu_bwd(size_t n, const float* y, const float* dy, float* dz, hipStream_t st) {
  if (n == 0) return TEMP_OK;
  if (n % 4) return TEMP_E_UNSUPPORTED;
  const size_t n4 = n / 4;
  int grid = ceil_div((long long)n4, 256);
  if (grid > 4096) grid = 4096;
  TEMP_LAUNCH(K_RELU_BWD, k_relu_bwd, dim3(grid), dim3(256), 0, st, n4, (const float4*)y, (const float4*)dy, (float4*)dz);
  return launch_status();
}

__global__ void __launch_bounds__(256) k_gather_rows(int n, int d4, const float4* __restrict__ table, const int32_t* __restrict__ idx,
                                                     float4* __restrict__ out) {
  const size_t total = (size_t)n * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / d4), c = (int)(i - (size_t)r * d4);
    const int s = idx[r];
    out[i] = (s >= 0) ? table[(size_t)s * d4 + c] : zero4();
  }
}

__global__ void __launch_bounds__(256) k_scatter_add_rows(int n, int d, const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                          float* __restrict__ table) {
  const size_t total = (size_t)n * d;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / d), c = (int)(i - (size_t)r * d);
    const int s = idx[r];
    if (s >= 0) atomicAdd(table + (size_t)s * d + c, src[i]);
  }
}

// out[s] = sum over j in [seg_ptr[s], seg_ptr[s+1]) of src[order[j]]  -- the adjoint of a row gather whose index
// list is known in advance (its inverse, grouped by table row, is built once on the host).  Deterministic
// replacement of the atomic scatter for hot tables: GDELT has 500 entities and ~100 k gathered rows per step,
// i.e. ~200 atomic adds per table element.  One wave per segment, one float4 per lane, 4 row loads in flight;
// the d/4-lane groups of a wave (LPR lanes each) take every (64/LPR)-th row and are summed by shuffles.
template <int LPR>
__global__ void __launch_bounds__(256) k_segment_sum_rows(int n_seg, int d4, const int32_t* __restrict__ seg_ptr,
                                                          const int32_t* __restrict__ order, const float4* __restrict__ src,
                                                          const int32_t* __restrict__ row_mask, const float4* __restrict__ relu_of,
                                                          float4* __restrict__ out) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lr = lane - grp * LPR;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool col_ok = lr < d4;
  for (int s = wave; s < n_seg; s += nwaves) {
    const int beg = seg_ptr[s], end = seg_ptr[s + 1];
    float4 acc = zero4();
    int j = beg + grp;
    for (; j + 3 * G < end; j += 4 * G) {
      const int r0 = order[j], r1 = order[j + G], r2 = order[j + 2 * G], r3 = order[j + 3 * G];
      const bool m0 = !row_mask || row_mask[r0] > 0, m1 = !row_mask || row_mask[r1] > 0, m2 = !row_mask || row_mask[r2] > 0,
                 m3 = !row_mask || row_mask[r3] > 0;          // masked rows were never written by their producer
      float4 v0 = zero4(), v1 = zero4(), v2 = zero4(), v3 = zero4();
      if (col_ok) {
        if (m0) v0 = src[(size_t)r0 * d4 + lr];
        if (m1) v1 = src[(size_t)r1 * d4 + lr];
        if (m2) v2 = src[(size_t)r2 * d4 + lr];
        if (m3) v3 = src[(size_t)r3 * d4 + lr];
      }
      acc = add4(add4(acc, v0), add4(v1, add4(v2, v3)));
    }
    for (; j < end; j += G) {
      const int r = order[j];
      if (col_ok && (!row_mask || row_mask[r] > 0)) acc = add4(acc, src[(size_t)r * d4 + lr]);
    }
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) acc = add4(acc, shfl_xor4(acc, m));
    if (grp == 0 && col_ok) out[(size_t)s * d4 + lr] = relu_of ? relu_gate4(relu_of[(size_t)s * d4 + lr], acc) : acc;
  }
}

// The same for segments of one or two rows (the adjoint of a gather whose rows are mostly distinct).
template <int LPR>
__global__ void __launch_bounds__(256) k_segment_sum_rows_short(int n_seg, int d4, const int32_t* __restrict__ seg_ptr,
                                                          const int32_t* __restrict__ order, const float4* __restrict__ src,
                                                          const int32_t* __restrict__ row_mask, const float4* __restrict__ relu_of,
                                                          float4* __restrict__ out) {
  // A wave takes FOUR consecutive segments at a time and walks them in lockstep: the three dependent round trips of a segment
  // (seg_ptr -> order -> row) are then shared by four segments instead of paid by each (the gather adjoints have 1-2 rows per
  // segment: the walk is all latency).
  constexpr int G = 64 / LPR, U = 4;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lr = lane - grp * LPR;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool col_ok = lr < d4;
  for (int s0 = wave * U; s0 < n_seg; s0 += nwaves * U) {
    int beg[U], len[U], maxlen = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = s0 + u < n_seg;
      beg[u] = ok ? seg_ptr[s0 + u] : 0;
      len[u] = ok ? seg_ptr[s0 + u + 1] - beg[u] : 0;
      maxlen = max(maxlen, len[u]);
    }
    float4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = zero4();
    for (int k = grp; k < maxlen; k += G) {
      int r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = k < len[u] ? order[beg[u] + k] : -1;
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v[u] = zero4();
        if (r[u] >= 0 && col_ok && (!row_mask || row_mask[r[u]] > 0)) v[u] = src[(size_t)r[u] * d4 + lr];   // masked rows were never written
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = add4(acc[u], v[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int m = LPR; m < 64; m <<= 1) acc[u] = add4(acc[u], shfl_xor4(acc[u], m));
      if (grp == 0 && col_ok && s0 + u < n_seg)
        out[(size_t)(s0 + u) * d4 + lr] = relu_of ? relu_gate4(relu_of[(size_t)(s0 + u) * d4 + lr], acc[u]) : acc[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Link-prediction loss over candidate lists (TKG_Module.train_link_prediction, models/TKG_Module.py:202-213):
// the scores of every positive against ALL entities come from one MFMA GEMM (query . all_embeds^T);
// these kernels pick the 1 + negative_rate candidates of each row out of that matrix and do the
// cross-entropy with label 0 -- nothing of shape (P, 1+neg, D) is ever materialised.
// One workgroup per row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce_256(float v, float* red, bool is_max) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();                                     // red may still be read from a previous reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// Candidate lists shorter than half a score row: gather the C logits once (<= 4 per thread in registers).
__global__ void __launch_bounds__(256) k_gather_ce_fwd(int C, int N, const float* __restrict__ scores, const int32_t* __restrict__ cand,
                                                       float* __restrict__ loss_rows, float* __restrict__ lse_rows) {
  __shared__ float red[4];
  const int p = blockIdx.x;
  const float* srow = scores + (size_t)p * N;
  const int32_t* crow = cand + (size_t)p * C;
  float mx = -INFINITY, sum = 0.f;
  if (C <= 1024) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = threadIdx.x + 256 * u;
      v[u] = k < C ? srow[crow[k]] : -INFINITY;
      mx = fmaxf(mx, v[u]);
    }
    mx = block_reduce_256(mx, red, true);
#pragma unroll
    for (int u = 0; u < 4; ++u) sum += expf(v[u] - mx);                 // exp(-inf) = 0 for the padding
  } else {
    for (int k = threadIdx.x; k < C; k += 256) mx = fmaxf(mx, srow[crow[k]]);
    mx = block_reduce_256(mx, red, true);
    for (int k = threadIdx.x; k < C; k += 256) sum += expf(srow[crow[k]] - mx);
  }
  sum = block_reduce_256(sum, red, false);
  if (threadIdx.x == 0) {
    const float lse = mx + logf(sum);
    lse_rows[p] = lse;
    loss_rows[p] = lse - srow[crow[0]];
  }
}

// Candidate lists about as long as the row (negative_rate ~ N_ents: the same entity is drawn several times): count the
// multiplicity of every entity with integer LDS atomics (order-independent), then ONE coalesced pass over the score row:
//   lse = log sum_e cnt[e] exp(s[e]).
__global__ void __launch_bounds__(256) k_gather_ce_fwd_cnt(int C, int N, const float* __restrict__ scores, const int32_t* __restrict__ cand,
                                                           float* __restrict__ loss_rows, float* __restrict__ lse_rows) {
  extern __shared__ int cnt[];
  __shared__ float red[4];
  const int p = blockIdx.x;
  for (int i = threadIdx.x; i < N; i += 256) cnt[i] = 0;
  __syncthreads();
  const float* srow = scores + (size_t)p * N;
  const int32_t* crow = cand + (size_t)p * C;
  for (int k = threadIdx.x; k < C; k += 256) atomicAdd(&cnt[crow[k]], 1);
  __syncthreads();
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256)
    if (cnt[i]) mx = fmaxf(mx, srow[i]);
  mx = block_reduce_256(mx, red, true);
  float sum = 0.f;
  for (int i = threadIdx.x; i < N; i += 256)
    if (cnt[i]) sum += (float)cnt[i] * expf(srow[i] - mx);
  sum = block_reduce_256(sum, red, false);
  if (threadIdx.x == 0) {
    const float lse = mx + logf(sum);
    lse_rows[p] = lse;
    loss_rows[p] = lse - srow[crow[0]];
  }
}

// The same for SHORT score rows (N <= 1024: GDELT's 500 entities under 48 000 loss rows): one WAVE per row, four rows per
// workgroup -- no workgroup barriers, no cross-wave reductions; a wave's LDS operations complete in issue order, so its zero fill,
// its integer atomics and its reads of the counters need no barrier between them.
__device__ __forceinline__ float wave_reduce_f(float v, bool is_max) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  return v;
}

__device__ __forceinline__ int* gather_ce_wave_counts(int* cnt_all, int N, int C, const int32_t* __restrict__ crow) {
  const int lane = threadIdx.x & 63;
  int* cnt = cnt_all + (threadIdx.x >> 6) * N;
  for (int i = lane; i < N; i += 64) cnt[i] = 0;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int k = lane; k < C; k += 64) atomicAdd(&cnt[crow[k]], 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return cnt;
}

__global__ void __launch_bounds__(256) k_gather_ce_fwd_cnt_w(int P, int C, int N, const float* __restrict__ scores, const int32_t* __restrict__ cand,
                                                             float* __restrict__ loss_rows, float* __restrict__ lse_rows) {
  extern __shared__ int cnt_all[];
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= P) return;
  const float* srow = scores + (size_t)p * N;
  const int32_t* crow = cand + (size_t)p * C;
  const int* cnt = gather_ce_wave_counts(cnt_all, N, C, crow);
  float mx = -INFINITY;
  float sv[16];                                                           // the row's scores of this lane (N <= 1024): read once
  int cv[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int i = lane + 64 * u;
    cv[u] = i < N ? cnt[i] : 0;
    sv[u] = i < N ? srow[i] : 0.f;
    if (cv[u]) mx = fmaxf(mx, sv[u]);
  }
  mx = wave_reduce_f(mx, true);
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if (cv[u]) sum += (float)cv[u] * expf(sv[u] - mx);
  sum = wave_reduce_f(sum, false);
  if (lane == 0) {
    const float lse = mx + logf(sum);
    lse_rows[p] = lse;
    loss_rows[p] = lse - srow[crow[0]];
  }
}

__global__ void __launch_bounds__(256) k_gather_ce_bwd_w(int P, int C, int N, const float* __restrict__ scores, const int32_t* __restrict__ cand,
                                                         const float* __restrict__ lse_rows, const float* __restrict__ scale_ptr, float inv_rows,
                                                         const float* __restrict__ row_scale, float* __restrict__ d_scores) {
  extern __shared__ int cnt_all[];
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= P) return;
  const float* srow = scores + (size_t)p * N;
  const int32_t* crow = cand + (size_t)p * C;
  const int* cnt = gather_ce_wave_counts(cnt_all, N, C, crow);
  const float lse = lse_rows[p];
  const float scale = scale_ptr[0] * (row_scale ? row_scale[p] : inv_rows);
  const int truth = crow[0];
  float* drow = d_scores + (size_t)p * N;
  for (int i = lane; i < N; i += 64) {
    const int c = cnt[i];
    float g = c ? (float)c * expf(srow[i] - lse) : 0.f;
    if (i == truth) g -= 1.f;
    drow[i] = g * scale;
  }
}

// d_scores[p, e] = scale * (cnt[e] * softmax(e) - [e == cand[p,0]])   (row written once, coalesced; multiplicities counted
// with integer LDS atomics, so the result does not depend on the order the candidates are visited in)
__global__ void __launch_bounds__(256) k_gather_ce_bwd(int C, int N, const float* __restrict__ scores, const int32_t* __restrict__ cand,
                                                       const float* __restrict__ lse_rows, const float* __restrict__ scale_ptr, float inv_rows,
                                                       const float* __restrict__ row_scale, float* __restrict__ d_scores) {
  extern __shared__ int cnt[];
  const int p = blockIdx.x;
  for (int i = threadIdx.x; i < N; i += 256) cnt[i] = 0;
  __syncthreads();
  const float* srow = scores + (size_t)p * N;
  const int32_t* crow = cand + (size_t)p * C;
  for (int k = threadIdx.x; k < C; k += 256) atomicAdd(&cnt[crow[k]], 1);
  __syncthreads();
  const float lse = lse_rows[p];
  const float scale = scale_ptr[0] * (row_scale ? row_scale[p] : inv_rows);
  const int truth = crow[0];
  float* drow = d_scores + (size_t)p * N;
  for (int i = threadIdx.x; i < N; i += 256) {
    const int c = cnt[i];
    float g = c ? (float)c * expf(srow[i] - lse) : 0.f;
    if (i == truth) g -= 1.f;
    drow[i] = g * scale;
  }
}

// ---------------------------------------------------------------------------------------------
// Folded query of the bilinear scorers (utils/scores.py:4-12 DistMult, :26-44 ComplEx), with the two row gathers
// fused in:  k = ent_rows[known_idx[p]],  r = rel[rel_idx[p]],
//   DistMult            q = k * r
//   ComplEx, tail mode  q = [re_k re_r - im_k im_r | re_k im_r + im_k re_r]      (k is the subject, candidates are objects)
//   ComplEx, head mode  q = [re_r re_k + im_r im_k | re_r im_k - im_r re_k]      (k is the object, candidates are subjects)
// so that score(candidate c) = <q, c>.  One thread per float4 of the half width; the backward writes the per-row
// gradients of k and r (the caller reduces them over the static index lists with temp_segment_sum_rows).
// ---------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void __launch_bounds__(256) k_bilinear_query(int P, int d, int kind, const float* __restrict__ ent_rows, const int32_t* __restrict__ known_idx,
                                                        const float* __restrict__ rel, const int32_t* __restrict__ rel_idx,
                                                        const int32_t* __restrict__ is_tail, const float* __restrict__ dq, float* __restrict__ o0,
                                                        float* __restrict__ o1) {
  const int half = kind == TEMP_SCORE_COMPLEX ? d / 2 : d;
  const int g4 = half / 4;
  const size_t total = (size_t)P * g4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(i / g4), j = (int)(i - (size_t)p * g4) * 4;
    const float* k = ent_rows + (size_t)known_idx[p] * d + j;
    const float* r = rel + (size_t)rel_idx[p] * d + j;
    const size_t o = (size_t)p * d + j;
    if (kind != TEMP_SCORE_COMPLEX) {
      const float4 kv = ld4(k), rv = ld4(r);
      if (!BWD) {
        st4(o0 + o, make_float4(kv.x * rv.x, kv.y * rv.y, kv.z * rv.z, kv.w * rv.w));
      } else {
        const float4 g = ld4(dq + o);
        st4(o0 + o, make_float4(g.x * rv.x, g.y * rv.y, g.z * rv.z, g.w * rv.w));
        st4(o1 + o, make_float4(g.x * kv.x, g.y * kv.y, g.z * kv.z, g.w * kv.w));
      }
      continue;
    }
    const float4 rk = ld4(k), ik = ld4(k + half), rr = ld4(r), ir = ld4(r + half);
    const float sg = is_tail[p] ? 1.f : -1.f;       // tail: q1 = rk rr - ik ir, q2 = rk ir + ik rr;  head: q1 = rk rr + ik ir, q2 = ik rr - rk ir
    if (!BWD) {
      st4(o0 + o, make_float4(rk.x * rr.x - sg * ik.x * ir.x, rk.y * rr.y - sg * ik.y * ir.y, rk.z * rr.z - sg * ik.z * ir.z, rk.w * rr.w - sg * ik.w * ir.w));
      st4(o0 + o + half, make_float4(ik.x * rr.x + sg * rk.x * ir.x, ik.y * rr.y + sg * rk.y * ir.y, ik.z * rr.z + sg * rk.z * ir.z, ik.w * rr.w + sg * rk.w * ir.w));
    } else {
      const float4 a = ld4(dq + o), b = ld4(dq + o + half);
      // q1 = rk rr - sg ik ir ; q2 = ik rr + sg rk ir
      st4(o0 + o, make_float4(a.x * rr.x + sg * b.x * ir.x, a.y * rr.y + sg * b.y * ir.y, a.z * rr.z + sg * b.z * ir.z, a.w * rr.w + sg * b.w * ir.w));                 // d re_k
      st4(o0 + o + half, make_float4(b.x * rr.x - sg * a.x * ir.x, b.y * rr.y - sg * a.y * ir.y, b.z * rr.z - sg * a.z * ir.z, b.w * rr.w - sg * a.w * ir.w));          // d im_k
      st4(o1 + o, make_float4(a.x * rk.x + b.x * ik.x, a.y * rk.y + b.y * ik.y, a.z * rk.z + b.z * ik.z, a.w * rk.w + b.w * ik.w));                                     // d re_r
      st4(o1 + o + half, make_float4(sg * (b.x * rk.x - a.x * ik.x), sg * (b.y * rk.y - a.y * ik.y), sg * (b.z * rk.z - a.z * ik.z), sg * (b.w * rk.w - a.w * ik.w)));  // d im_r
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Snapshot store: the edge views of a batch (a disjoint union of member snapshots whose sorted / chunked views are
// resident on the device) are the members' arrays back to back with per-member offsets added.  One launch copies every
// array of every member: a descriptor names a member array, its place in the packed output and how to shift it;
// a piece is up to TEMP_ASSEMBLE_PIECE elements of one descriptor (one workgroup each).
//   mode 0: out = v + add        mode 1: out = v >= 0 ? v + add : v   (partial-sum slots, -1 = none)
//   mode 2: t = table[add + aux[i]];  out = t >= 0 ? t + v : t        (by-relation slots: base of the member x relation + rank)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_assemble_views(const int32_t* __restrict__ piece_desc, const int32_t* __restrict__ piece_start,
                                                        const TempCopyDesc* __restrict__ descs, const int32_t* __restrict__ table,
                                                        int32_t* __restrict__ out) {
  const TempCopyDesc d = descs[piece_desc[blockIdx.x]];
  const int start = piece_start[blockIdx.x];
  const int end = min(d.len, start + TEMP_ASSEMBLE_PIECE);
  int32_t* __restrict__ o = out + d.dst_off;
  for (int i = start + threadIdx.x; i < end; i += 256) {
    const int v = d.src[i];
    int r;
    if (d.mode == 0) r = v + d.add;
    else if (d.mode == 1) r = v >= 0 ? v + d.add : v;
    else { const int t = table[d.add + d.aux[i]]; r = t >= 0 ? t + v : t; }
    o[i] = r;
  }
}

// ---------------------------------------------------------------------------------------------
// Filtered negative sampling (CorruptTriples.negative_sampling / corrupt_triple, utils/CorrptTriples.py:36-85):
// for every positive row, K corrupted entities drawn uniformly over ALL entities, redrawing those that form a true
// triple of the target snapshot (the row's known-true set is the slice ids[lo[row] .. hi[row]) of a resident store).
// One thread per candidate; the draw is a counter-based hash of (seed, row, column, attempt), so a step's samples
// are a pure function of its seed (no generator state, no rejection ROUNDS over the whole matrix).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256) k_corrupt_sample(long long total, int K1, int N, unsigned long long seed, const int32_t* __restrict__ truth,
                                                        const int32_t* __restrict__ lo, const int32_t* __restrict__ hi,
                                                        const int32_t* __restrict__ ids, int32_t* __restrict__ cand) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / K1), k = (int)(i - (long long)row * K1);
    if (k == 0) { cand[i] = truth[row]; continue; }
    const int a = lo ? lo[row] : 0, b = lo ? hi[row] : 0;
    const unsigned long long base = splitmix64(seed ^ ((unsigned long long)row * 0xD1B54A32D192ED03ull + (unsigned long long)k));
    const int len = b - a;
    int c = 0;
    bool done = false;
    if (len > 16 && len < N) {                        // long known-true set: rejection with a binary search per attempt
      for (int attempt = 0; attempt < 64 && !done; ++attempt) {
        const unsigned long long x = splitmix64(base + attempt);
        c = (int)(((x >> 32) * (unsigned long long)N) >> 32);
        int l = a, h = b;
        while (l < h) { const int m = (l + h) >> 1; if (ids[m] < c) l = m + 1; else h = m; }
        done = !(l < b && ids[l] == c);
      }
    }
    if (!done) {
      // exact: the u-th entity of the complement, u uniform in [0, N - len) -- walk the ascending list, skipping its members
      const int free_n = len < N ? N - len : N;       // nothing allowed (the reference would loop forever): plain uniform draw
      const unsigned long long x = splitmix64(base + 64);
      c = (int)(((x >> 32) * (unsigned long long)free_n) >> 32);
      if (len < N)
        for (int j = a; j < b && ids[j] <= c; ++j) ++c;
    }
    cand[i] = c;
  }
}

// ---------------------------------------------------------------------------------------------
// Filtered rank of one test triple per workgroup (utils/evaluation.py:40-106): the reference sets the scores of the
// other known-true entities to -10e6, applies a sigmoid and takes the target's position in a descending sort.
// Position in a STABLE descending order = #(strictly larger) + #(equal with a smaller entity id) + 1, so nothing is
// sorted: one pass over the score row counts, a second pass over the row's filter list replaces the contribution of
// each filtered entity by that of sigmoid(-10e6) = 0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rank_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ int rank_ahead(float v, int j, float ts, int tgt) { return (v > ts) | ((v == ts) & (j < tgt)); }

__global__ void __launch_bounds__(256) k_filtered_rank(int N, int ld, const float* __restrict__ scores, const int32_t* __restrict__ target,
                                                       const int32_t* __restrict__ filt_ptr, const int32_t* __restrict__ filt_ids,
                                                       int32_t* __restrict__ ranks) {
  __shared__ int red[4];
  const int p = blockIdx.x;
  const float* srow = scores + (size_t)p * ld;
  const int tgt = target[p];
  const float ts = rank_sigmoid(srow[tgt]);
  int cnt = 0;
  const int n4 = N & ~3;
  for (int j = threadIdx.x * 4; j < n4; j += 1024) {
    const float4 s = *reinterpret_cast<const float4*>(srow + j);
    cnt += rank_ahead(rank_sigmoid(s.x), j, ts, tgt) + rank_ahead(rank_sigmoid(s.y), j + 1, ts, tgt)
         + rank_ahead(rank_sigmoid(s.z), j + 2, ts, tgt) + rank_ahead(rank_sigmoid(s.w), j + 3, ts, tgt);
  }
  for (int j = n4 + threadIdx.x; j < N; j += 256) cnt += rank_ahead(rank_sigmoid(srow[j]), j, ts, tgt);
  if (filt_ptr) {
    for (int f = filt_ptr[p] + threadIdx.x; f < filt_ptr[p + 1]; f += 256) {
      const int j = filt_ids[f];
      if (j == tgt) continue;
      cnt += rank_ahead(0.0f, j, ts, tgt) - rank_ahead(rank_sigmoid(srow[j]), j, ts, tgt);
    }
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) ranks[p] = red[0] + red[1] + red[2] + red[3] + 1;
}

__global__ void __launch_bounds__(256) k_copy(size_t n16, const float4* __restrict__ src, float4* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---- trace state -------------------------------------------------------------------------------
struct TraceState { hipEvent_t* ev; int* ids; int cap; int n; };
static TraceState* g_trace = nullptr;

int trace_open(int kernel_id, hipStream_t st) {
  TraceState* t = g_trace;
  if (!t) return -1;
  const int slot = __atomic_fetch_add(&t->n, 1, __ATOMIC_RELAXED);
  if (slot >= t->cap) return -1;
  t->ids[slot] = kernel_id;
  (void)hipEventRecord(t->ev[2 * slot], st);
  return slot;
}
void trace_close(int slot, hipStream_t st) {
  TraceState* t = g_trace;
  if (!t || slot < 0) return;
  (void)hipEventRecord(t->ev[2 * slot + 1], st);
}

}  // namespace temp

using namespace temp;

extern "C" {

int temp_trace_begin(int capacity) {
  if (capacity <= 0 || g_trace) return TEMP_E_BADARG;
  TraceState* t = new TraceState();
  t->cap = capacity;
  t->n = 0;
  t->ids = new int[capacity];
  t->ev = new hipEvent_t[2 * (size_t)capacity];
  for (int i = 0; i < 2 * capacity; ++i)
    if (hipEventCreate(&t->ev[i]) != hipSuccess) return TEMP_E_LAUNCH;
  g_trace = t;
  return TEMP_OK;
}

int temp_trace_end(int* kernel_ids, float* ms, int capacity, int* n_out) {
  TraceState* t = g_trace;
  if (!t || !n_out) return TEMP_E_BADARG;
  g_trace = nullptr;
  if (hipDeviceSynchronize() != hipSuccess) return TEMP_E_LAUNCH;
  int n = t->n < t->cap ? t->n : t->cap;
  if (n > capacity) n = capacity;
  for (int i = 0; i < n; ++i) {
    float v = 0.f;
    (void)hipEventElapsedTime(&v, t->ev[2 * i], t->ev[2 * i + 1]);
    if (kernel_ids) kernel_ids[i] = t->ids[i];
    if (ms) ms[i] = v;
  }
  *n_out = n;
  for (int i = 0; i < 2 * t->cap; ++i) (void)hipEventDestroy(t->ev[i]);
  delete[] t->ev;
  delete[] t->ids;
  delete t;
  return TEMP_OK;
}

}  // extern "C"

// ---- scratch slots for the packed (bf16-split) weight matrices of gemm_bx.hpp and the k-slice partial products: static device
// memory (one copy per device: a __device__ array is instantiated on every device that loads the module), BX_SLOTS slots of
// BX_SLOT_BYTES.  A slot belongs to one (device, stream): two launches on one stream are ordered, so its slot is free again when
// the next pack kernel of that stream runs.  With every slot taken, the least recently used one is handed to the new stream --
// safe when the previous owner's last launch has drained, which holds for streams that were destroyed or went idle (the callers'
// pools keep <= BX_SLOTS streams live per device; beyond that two live streams could share a slot, so the hand-over first checks
// hipStreamQuery of the old owner and refuses (nullptr: the callers fall back to the kernels that need no scratch, counted in
// g_bx_refused) while it still has work in flight).  Nothing is allocated, freed or synchronised at run time.
__device__ __attribute__((aligned(16))) unsigned char g_bx_slots[BX_SLOTS][BX_SLOT_BYTES];

namespace temp {
static std::atomic<long long> g_bx_refused{0};
long long bx_scratch_refused() { return g_bx_refused.load(std::memory_order_relaxed); }

bx_u32x4* bx_scratch(hipStream_t st, size_t bytes) {
  constexpr int MAX_DEV = 16;
  struct DevSlots { unsigned char* base = nullptr; hipStream_t owner[BX_SLOTS] = {}; unsigned long long used[BX_SLOTS] = {}; bool pinned[BX_SLOTS] = {}; int n = 0; };
  static std::mutex mu;
  static DevSlots devs[MAX_DEV];
  static unsigned long long tick = 0;
  if (bytes > BX_SLOT_BYTES) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) { g_bx_refused.fetch_add(1, std::memory_order_relaxed); return nullptr; }
  std::lock_guard<std::mutex> lock(mu);
  DevSlots& d = devs[dev];
  if (!d.base) {
    void* p = nullptr;                                        // the symbol's address on the CURRENT device
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_bx_slots)) != hipSuccess || !p) { g_bx_refused.fetch_add(1, std::memory_order_relaxed); return nullptr; }
    d.base = (unsigned char*)p;
  }
  int slot = -1;
  for (int i = 0; i < d.n; ++i)
    if (d.owner[i] == st) { slot = i; break; }
  if (slot < 0) {
    if (d.n < BX_SLOTS) {
      slot = d.n++;
    } else {
      // Recycle the least recently used slot whose owner has drained -- but never one that was handed out under a HIP-graph
      // capture: the captured graph has the slot's address baked in and may be replayed at any time, on any stream, while its
      // capture stream sits idle.  (hipStreamQuery on a capturing stream would also invalidate that capture.)
      int lru = -1;
      for (int i = 0; i < BX_SLOTS; ++i)
        if (!d.pinned[i] && (lru < 0 || d.used[i] < d.used[lru])) lru = i;
      if (lru < 0) { g_bx_refused.fetch_add(1, std::memory_order_relaxed); return nullptr; }
      // the old owner may be a destroyed stream (query fails: its work has drained) or an idle one (hipSuccess)
      const hipError_t q = hipStreamQuery(d.owner[lru]);
      if (q == hipErrorNotReady) { g_bx_refused.fetch_add(1, std::memory_order_relaxed); return nullptr; }
      (void)hipGetLastError();
      slot = lru;
    }
    d.owner[slot] = st;
  }
  if (!d.pinned[slot]) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) d.pinned[slot] = true;
    else (void)hipGetLastError();
  }
  d.used[slot] = ++tick;
  return reinterpret_cast<bx_u32x4*>(d.base + (size_t)slot * BX_SLOT_BYTES);
}

// ---- kernel-selection switches (include/temp_amd.h).  Defaults, then the environment, once at load time.
static std::atomic<int> g_options[TEMP_OPT_COUNT];
static const bool g_options_init = [] {
  g_options[TEMP_OPT_MFMA_BF16X3] = 1; g_options[TEMP_OPT_TN_SPLIT] = 1; g_options[TEMP_OPT_RGCN_SCALAR] = 1;
  g_options[TEMP_OPT_GEMM_STREAM] = 0; g_options[TEMP_OPT_GRU_STREAM] = 0; g_options[TEMP_OPT_RGCN_TILE] = 1; g_options[TEMP_OPT_DEBUG] = 0; g_options[TEMP_OPT_OVERLAP] = 1; g_options[TEMP_OPT_GEMM_RESIDENT] = 1;
  g_options[TEMP_OPT_MFMA_F16X2] = 1;
  const char* e;
  if ((e = getenv("TEMP_MFMA")) && e[0] == 'f' && e[1] == '3') g_options[TEMP_OPT_MFMA_BF16X3] = 0;      // f32
  if ((e = getenv("TEMP_MFMA")) && e[0] == 'b') g_options[TEMP_OPT_MFMA_F16X2] = 0;                       // bf16x3
  if ((e = getenv("TEMP_TN_SPLIT")) && e[0] == '0') g_options[TEMP_OPT_TN_SPLIT] = 0;
  if ((e = getenv("TEMP_RGCN_SCALAR")) && e[0] == '0') g_options[TEMP_OPT_RGCN_SCALAR] = 0;
  if ((e = getenv("TEMP_GEMM_STREAM")) && e[0] == '1') g_options[TEMP_OPT_GEMM_STREAM] = 1;
  if ((e = getenv("TEMP_GRU_STREAM")) && e[0] == '1') g_options[TEMP_OPT_GRU_STREAM] = 1;
  if ((e = getenv("TEMP_RGCN_TILE")) && e[0] >= '0' && e[0] <= '9') g_options[TEMP_OPT_RGCN_TILE] = atoi(e);
  if ((e = getenv("TEMP_OVERLAP")) && e[0] == '0') g_options[TEMP_OPT_OVERLAP] = 0;
  if ((e = getenv("TEMP_GEMM_RESIDENT")) && e[0] == '0') g_options[TEMP_OPT_GEMM_RESIDENT] = 0;
  if ((e = getenv("TEMP_DEBUG"))) g_options[TEMP_OPT_DEBUG] = atoi(e);
  return true;
}();
int option(int key) { return (key >= 0 && key < TEMP_OPT_COUNT) ? g_options[key].load(std::memory_order_relaxed) : -1; }
static std::atomic<long long> g_hx_launches{0};
void hx_count() { g_hx_launches.fetch_add(1, std::memory_order_relaxed); }
long long hx_launches() { return g_hx_launches.load(std::memory_order_relaxed); }
}  // namespace temp

extern "C" {

const char* temp_trace_kernel_name(int id) {
  static const char* names[] = {"k_rgcn_agg<fwd>", "k_rgcn_agg<dx>", "k_rgcn_dw", "k_fixup", "k_gemm_panel<loop_fwd>",
                                "k_gemm_panel<loop_dx>", "k_gemm_tn", "k_reduce_slices", "k_colsum_part", "k_relu_bwd", "k_gru_fwd",
                                "k_gru_bwd_gates", "k_gemm_panel<gru_dx>", "k_gemm_panel<gru_dprev>", "k_gather_rows",
                                "k_scatter_add_rows", "k_decay_grad", "k_copy", "k_gemm_panel<isolated>", "k_gemm_panel<gru_gi>",
                                "k_gemm_panel<linear>", "k_gather_ce", "k_sa_attn_fwd", "k_sa_attn_bwd", "k_gru_chain_fwd", "k_gru_chain_bwd",
                                "k_gru_chain_pack", "k_bx_pack", "k_gemm_tn_bx8", "k_gemm_tn_bx", "k_gru_wgrad", "k_segment_sum_rows", "k_absmax_keys"};
  return (id >= 0 && id < K_COUNT) ? names[id] : "?";
}

int temp_abi_version(void) { return TEMP_ABI_VERSION; }

int temp_set_option(int key, int value) {
  if (key < 0 || key >= TEMP_OPT_COUNT) return -1;
  return temp::g_options[key].exchange(value, std::memory_order_relaxed);
}
int temp_get_option(int key) { return temp::option(key); }
long long temp_scratch_refused(void) { return temp::bx_scratch_refused(); }
long long temp_f16_launches(void) { return temp::hx_launches(); }

const char* temp_error_string(int code) {
  switch (code) {
    case TEMP_OK: return "ok";
    case TEMP_E_BADARG: return "bad argument (NULL, negative or inconsistent)";
    case TEMP_E_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case TEMP_E_WORKSPACE: return "workspace missing or too small";
    case TEMP_E_LAUNCH: return "HIP launch failure";
    default: return "unknown error";
  }
}

int temp_rgcn_isolated_fwd(int n, int d, const float* e, const float* loop_w, const float* bias, int act, float* out, const TempDropout* drop,
                           void* stream) {
  if (n < 0 || d <= 0 || !loop_w || (n > 0 && (!e || !out))) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  const DropSpec ds = drop_spec(drop);
  return gemm_add_bias_act(K_GEMM_ISO, n, d, d, e, d, nullptr, loop_w, d, 0, e, d, nullptr, bias, act, out, d, (hipStream_t)stream, &ds);
}

size_t temp_rgcn_isolated_bwd_workspace(int n, int d) {
  if (n < 0 || d <= 0) return 0;
  return 2 * align_up((size_t)n * d * sizeof(float), 256) + gemm_tn_workspace(n, d, d) + colsum_workspace(n, d) + 256;
}

int temp_rgcn_isolated_bwd(int n, int d, const float* e, const float* out, const float* d_out_grad, const float* loop_w, int has_bias,
                           int act, float* d_e, float* d_loop_w, float* d_bias, void* workspace, size_t workspace_bytes, const TempDropout* drop,
                           void* stream) {
  if (n < 0 || d <= 0 || !loop_w || !d_loop_w || (n > 0 && (!e || !d_out_grad || !d_e))) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (act == TEMP_ACT_RELU && !out) return TEMP_E_BADARG;
  if (has_bias && !d_bias) return TEMP_E_BADARG;
  if (!workspace || workspace_bytes < temp_rgcn_isolated_bwd_workspace(n, d)) return TEMP_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  float* dzbuf = (float*)base;
  base += align_up((size_t)n * d * sizeof(float), 256);
  float* dzm_buf = (float*)base;
  base += align_up((size_t)n * d * sizeof(float), 256);
  void* tn = base;
  const size_t tnb = gemm_tn_workspace(n, d, d);
  base += tnb;
  void* cs = base;
  const size_t csb = colsum_workspace(n, d);
  const float* dz = d_out_grad;
  int rc;
  if (act == TEMP_ACT_RELU) {
    rc = relu_bwd((size_t)n * d, out, d_out_grad, dzbuf, st);
    if (rc) return rc;
    dz = dzbuf;
  }
  // d_e = dz + dzm . loop_w^T,  d_loop_w = e^T . dzm   (dzm = dz masked like the forward loop message; = dz without dropout)
  const DropSpec ds = drop_spec(drop);
  const float* dzm = dz;
  if (ds.p > 0.f) {
    rc = mask_rows(n, d, dz, dzm_buf, ds, st);
    if (rc) return rc;
    dzm = dzm_buf;
  }
  rc = gemm_add_bias_act(K_GEMM_ISO, n, d, d, dzm, d, nullptr, loop_w, d, 1, dz, d, nullptr, nullptr, TEMP_ACT_NONE, d_e, d, st);
  if (rc) return rc;
  rc = gemm_tn(n, d, d, e, d, dzm, d, d_loop_w, d, tn, tnb, st);
  if (rc) return rc;
  if (has_bias) rc = colsum(n, d, dz, d, d_bias, cs, csb, st);
  return rc;
}

int temp_gather_rows(int n, int d, const float* table, const int32_t* idx, float* out, void* stream) {
  if (n < 0 || d <= 0 || (n > 0 && (!table || !idx || !out))) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (n == 0) return TEMP_OK;
  int grid = ceil_div((long long)n * (d / 4), 256);
  if (grid > 4096) grid = 4096;
  TEMP_LAUNCH(K_GATHER_ROWS, k_gather_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, d / 4, (const float4*)table, idx, (float4*)out);
  return launch_status();
}

int temp_scatter_add_rows(int n, int d, const float* src, const int32_t* idx, float* table, void* stream) {
  if (n < 0 || d <= 0 || (n > 0 && (!table || !idx || !src))) return TEMP_E_BADARG;
  if (n == 0) return TEMP_OK;
  int grid = ceil_div((long long)n * d, 256);
  if (grid > 4096) grid = 4096;
  TEMP_LAUNCH(K_SCATTER_ADD, k_scatter_add_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, d, src, idx, table);
  return launch_status();
}

}  // extern "C"
namespace temp {
// Long segments (a hot table: hundreds of gathered rows per table row): one BLOCK per segment, its 4 waves take every
// 4th row with 8 row loads in flight each, partial sums meet in LDS in a fixed order.
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_segment_sum_rows_blk(int n_seg, int d4, const int32_t* __restrict__ seg_ptr,
                                                                     const int32_t* __restrict__ order, const float4* __restrict__ src,
                                                                     const int32_t* __restrict__ row_mask, const float4* __restrict__ relu_of,
                                                                     float4* __restrict__ out) {
  // one block per segment: wave w takes rows w, w + WAVES, ... eight at a time; the waves' sums are added in wave order.
  // WAVES = 16 for segments of a hundred rows and more (500 entities gathered 82 000 times: 164 rows each -- four waves walk
  // them in five dependent round trips of order[] -> row, sixteen in two)
  __shared__ float4 red[WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool col_ok = lane < d4;
  for (int s = blockIdx.x; s < n_seg; s += gridDim.x) {
    const int beg = seg_ptr[s], end = seg_ptr[s + 1];
    float4 acc = zero4();
    for (int j0 = beg + wave; j0 < end; j0 += 8 * WAVES) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + WAVES * u;
        v[u] = zero4();
        if (j < end && col_ok) {
          const int r = order[j];
          if (!row_mask || row_mask[r] > 0) v[u] = src[(size_t)r * d4 + lane];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && col_ok) {
      float4 t = red[0][lane];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) t = add4(t, red[w][lane]);
      out[(size_t)s * d4 + lane] = relu_of ? relu_gate4(relu_of[(size_t)s * d4 + lane], t) : t;
    }
    __syncthreads();
  }
}

// Two sources over the SAME segmentation in one launch (the table layer's backward sums the aggregation part of d_h and dz per
// table row: same inverse map, one walk of order[] instead of two; a stays masked by row_mask as in the single-source kernels).
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_segment_sum_rows_blk2(int n_seg, int d4a, int d4b, const int32_t* __restrict__ seg_ptr,
                                                                      const int32_t* __restrict__ order, const float4* __restrict__ src_a,
                                                                      const int32_t* __restrict__ mask_a, const float4* __restrict__ src_b,
                                                                      float4* __restrict__ out_a, float4* __restrict__ out_b) {
  __shared__ float4 red[2][WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool a_ok = lane < d4a, b_ok = lane < d4b;
  for (int s = blockIdx.x; s < n_seg; s += gridDim.x) {
    const int beg = seg_ptr[s], end = seg_ptr[s + 1];
    float4 acc_a = zero4(), acc_b = zero4();
    for (int j0 = beg + wave; j0 < end; j0 += 4 * WAVES) {
      float4 va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + WAVES * u;
        va[u] = zero4(); vb[u] = zero4();
        if (j < end) {
          const int r = order[j];
          if (a_ok && (!mask_a || mask_a[r] > 0)) va[u] = src_a[(size_t)r * d4a + lane];
          if (b_ok) vb[u] = src_b[(size_t)r * d4b + lane];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc_a = add4(acc_a, va[u]); acc_b = add4(acc_b, vb[u]); }
    }
    red[0][wave][lane] = acc_a;
    red[1][wave][lane] = acc_b;
    __syncthreads();
    if (wave < 2) {
      const bool ok = wave == 0 ? a_ok : b_ok;
      if (ok) {
        float4 t = red[wave][0][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t = add4(t, red[wave][w][lane]);
        if (wave == 0) out_a[(size_t)s * d4a + lane] = t; else out_b[(size_t)s * d4b + lane] = t;
      }
    }
    __syncthreads();
  }
}

int segment_sum_rows2(int n_seg, const int32_t* seg_ptr, const int32_t* order, int d_a, const float* src_a, const int32_t* mask_a, float* out_a,
                      int d_b, const float* src_b, float* out_b, hipStream_t st, long long n_rows_hint) {
  if (d_a % 4 == 0 && d_b % 4 == 0 && d_a <= 256 && d_b <= 256 && n_rows_hint >= 96LL * n_seg) {
    TEMP_LAUNCH(K_SEGMENT_SUM, k_segment_sum_rows_blk2<16>, dim3(n_seg < 4096 ? n_seg : 4096), dim3(16 * 64), 0, st, n_seg, d_a / 4, d_b / 4, seg_ptr, order,
                (const float4*)src_a, mask_a, (const float4*)src_b, (float4*)out_a, (float4*)out_b);
    return launch_status();
  }
  int rc = segment_sum_rows(n_seg, d_a, seg_ptr, order, src_a, mask_a, out_a, st, n_rows_hint);
  if (rc) return rc;
  return segment_sum_rows(n_seg, d_b, seg_ptr, order, src_b, nullptr, out_b, st, n_rows_hint);
}

// Very long segments (a 40-row relation table gathered 48 000 times by the loss): every segment is cut into S equal
// parts, one block per part writes its partial sum into the workspace, a second kernel adds the S partials in order.
__global__ void __launch_bounds__(256) k_segment_sum_part(int S, int d4, const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ order,
                                                          const float4* __restrict__ src, const int32_t* __restrict__ row_mask,
                                                          float4* __restrict__ part) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool col_ok = lane < d4;
  const int s = blockIdx.y, p = blockIdx.x;
  const int beg0 = seg_ptr[s], end0 = seg_ptr[s + 1];
  const int chunk = (end0 - beg0 + S - 1) / S;
  const int beg = beg0 + p * chunk, end = min(end0, beg + chunk);
  float4 acc = zero4();
  for (int j0 = beg + wave; j0 < end; j0 += 32) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + 4 * u;
      v[u] = zero4();
      if (j < end && col_ok) {
        const int r = order[j];
        if (!row_mask || row_mask[r] > 0) v[u] = src[(size_t)r * d4 + lane];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col_ok) part[((size_t)s * S + p) * d4 + lane] = add4(add4(red[0][lane], red[1][lane]), add4(red[2][lane], red[3][lane]));
}

__global__ void __launch_bounds__(256) k_segment_sum_fin(int n_seg, int S, int d4, const float4* __restrict__ part, const float4* __restrict__ relu_of,
                                                         float4* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_seg * d4) return;
  const size_t s = i / d4, c = i - s * d4;
  float4 acc = zero4();
  for (int p = 0; p < S; ++p) acc = add4(acc, part[(s * S + p) * d4 + c]);
  out[i] = relu_of ? relu_gate4(relu_of[i], acc) : acc;
}

// Segment sum over FIXED PIECES of the row list (skewed segmentations: the adjoint of a gather of Zipf-distributed entity rows has
// a few segments of hundreds to thousands of rows among thousands of short ones, and a wave per segment takes as long as the
// longest).  Wave c sums the rows order[32 c .. 32 c + 32) segment by segment, in row order -- all 32 row loads are issued before
// the first addition: one memory round trip per piece -- a segment that lies inside the piece is written to `out`; the FIRST and
// the LAST segment of the piece, when they reach beyond it, go to part[c][0] / part[c][1].  k_segment_sum_pieces_fin then adds the
// pieces of every such segment in piece order (and zero-fills the empty segments): fixed pieces, fixed order => bit-repeatable.
// Lanes = float4 columns (d4 <= 64).
#define SEGSUM_PIECE 32
#define SEGSUM_PIECE_LOG2 5
__global__ void __launch_bounds__(256) k_segment_sum_pieces(int n_seg, int n_rows, int d4, const int32_t* __restrict__ seg_ptr,
                                                            const int32_t* __restrict__ order, const float4* __restrict__ src,
                                                            const int32_t* __restrict__ row_mask, const float4* __restrict__ relu_of,
                                                            float4* __restrict__ out, float4* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int r0 = c * SEGSUM_PIECE;
  n_rows = min(n_rows, seg_ptr[n_seg]);                                    // (the rows there are: order[] holds exactly seg_ptr[n_seg])
  if (r0 >= n_rows) return;
  const int r1 = min(r0 + SEGSUM_PIECE, n_rows);
  const bool col_ok = lane < d4;
  const int col = col_ok ? lane : 0;
  int mine = (r0 + lane < r1) ? order[r0 + lane] : -1;
  if (row_mask && mine >= 0 && row_mask[mine] <= 0) mine = -1;             // masked rows were never written by their producer
  float4 v[SEGSUM_PIECE];
#pragma unroll
  for (int u = 0; u < SEGSUM_PIECE; ++u) {
    const int r = __builtin_amdgcn_readlane(mine, u);
    v[u] = r >= 0 ? src[(size_t)r * d4 + col] : zero4();
  }
  // the segment of row r0 (wave-uniform binary search: seg_ptr is non-decreasing)
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (seg_ptr[mid + 1] > r0) hi = mid; else lo = mid + 1;
  }
  int sg = lo, sb = seg_ptr[sg], se = seg_ptr[sg + 1];
  float4 acc = zero4();
  auto flush = [&]() {
    if (!col_ok) return;
    if (sb >= r0 && se <= r1) out[(size_t)sg * d4 + lane] = relu_of ? relu_gate4(relu_of[(size_t)sg * d4 + lane], acc) : acc;
    else part[((size_t)c * 2 + (sb <= r0 ? 0 : 1)) * d4 + lane] = acc;
  };
#pragma unroll
  for (int u = 0; u < SEGSUM_PIECE; ++u) {
    const int row = r0 + u;
    if (row < r1) {
      if (row == se) {                                                     // (wave-uniform) the next non-empty segment starts here
        flush();
        do { ++sg; se = seg_ptr[sg + 1]; } while (se <= row);
        sb = seg_ptr[sg];
        acc = zero4();
      }
      acc = add4(acc, v[u]);
    }
  }
  flush();
}

__global__ void __launch_bounds__(256) k_segment_sum_pieces_fin(int n_seg, int d4, const int32_t* __restrict__ seg_ptr, const float4* __restrict__ part,
                                                                const float4* __restrict__ relu_of, float4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const bool col_ok = lane < d4;
  for (int s = wave; s < n_seg; s += nwaves) {
    const int beg = seg_ptr[s], end = seg_ptr[s + 1];
    if (end <= beg) {
      if (col_ok) out[(size_t)s * d4 + lane] = zero4();
      continue;
    }
    const int cb = beg >> SEGSUM_PIECE_LOG2, ce = (end - 1) >> SEGSUM_PIECE_LOG2;
    if (cb == ce) continue;                                                // inside one piece: written by the walk
    float4 acc = zero4();
    if (col_ok) {
      acc = part[((size_t)cb * 2 + ((beg & (SEGSUM_PIECE - 1)) == 0 ? 0 : 1)) * d4 + lane];  // first piece: its last segment, unless it starts the piece
      int c = cb + 1;
      for (; c + 16 <= ce + 1; c += 16) {
        float4 q[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) q[u] = part[(size_t)(c + u) * 2 * d4 + lane];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = add4(acc, q[u]);
      }
      for (; c <= ce; ++c) acc = add4(acc, part[(size_t)c * 2 * d4 + lane]);
      out[(size_t)s * d4 + lane] = relu_of ? relu_gate4(relu_of[(size_t)s * d4 + lane], acc) : acc;
    }
  }
}

// rows per segment between the short-segment kernel's and the block-per-segment kernels' ranges: fixed pieces (robust to skew)
static bool segsum_pieces(int n_seg, long long n_rows, int d4) { return d4 <= 64 && n_rows > 2LL * n_seg && n_rows <= 32LL * n_seg && n_rows < (1LL << 31) - 64; }

static int segsum_splits(int n_seg, long long n_rows) {
  if (n_seg <= 0 || n_rows < 512LL * n_seg) return 1;
  int S = 2048 / n_seg;
  if (S > 64) S = 64;
  return S < 2 ? 1 : S;
}

size_t segment_sum_rows_workspace(int n_seg, long long n_rows, int d) {
  const int S = segsum_splits(n_seg, n_rows);
  if (S > 1) return (size_t)n_seg * S * d * sizeof(float);
  if (segsum_pieces(n_seg, n_rows, d / 4)) return (size_t)ceil_div(n_rows, (long long)SEGSUM_PIECE) * 2 * d * sizeof(float);
  return 0;
}

int segment_sum_rows(int n_seg, int d, const int32_t* seg_ptr, const int32_t* order, const float* src, const int32_t* row_mask, float* out,
                     hipStream_t st, long long n_rows_hint, void* ws, size_t ws_bytes, const float* relu_src) {
  const int d4 = d / 4;
  const float4* relu_of = (const float4*)relu_src;           // out = (relu_src > 0) ? sum : 0, element by element (nullable)
  const int S = segsum_splits(n_seg, n_rows_hint);
  if (S > 1 && d4 <= 64 && ws && ws_bytes >= segment_sum_rows_workspace(n_seg, n_rows_hint, d)) {
    TEMP_LAUNCH(K_SEGMENT_SUM, k_segment_sum_part, dim3(S, n_seg), dim3(256), 0, st, S, d4, seg_ptr, order, (const float4*)src, row_mask, (float4*)ws);
    TEMP_LAUNCH(K_SEGMENT_SUM, k_segment_sum_fin, dim3(ceil_div((long long)n_seg * d4, 256)), dim3(256), 0, st, n_seg, S, d4, (const float4*)ws, relu_of, (float4*)out);
    return launch_status();
  }
  if (S <= 1 && segsum_pieces(n_seg, n_rows_hint, d4) && ws && ws_bytes >= segment_sum_rows_workspace(n_seg, n_rows_hint, d)) {
    const int n_pieces = (int)ceil_div(n_rows_hint, (long long)SEGSUM_PIECE);
    TEMP_LAUNCH(K_SEGMENT_SUM, k_segment_sum_pieces, dim3(ceil_div(n_pieces, 4)), dim3(256), 0, st, n_seg, (int)n_rows_hint, d4, seg_ptr, order,
                (const float4*)src, row_mask, relu_of, (float4*)out, (float4*)ws);
    int grid = ceil_div(n_seg, 4);
    if (grid > 2048) grid = 2048;
    TEMP_LAUNCH(K_SEGMENT_SUM, k_segment_sum_pieces_fin, dim3(grid), dim3(256), 0, st, n_seg, d4, seg_ptr, (const float4*)ws, relu_of, (float4*)out);
    return launch_status();
  }
  if (n_rows_hint > 32LL * n_seg && d4 <= 64) {
    if (n_rows_hint >= 96LL * n_seg)
      TEMP_LAUNCH(K_SEGMENT_SUM, k_segment_sum_rows_blk<16>, dim3(n_seg < 4096 ? n_seg : 4096), dim3(16 * 64), 0, st, n_seg, d4, seg_ptr, order,
                  (const float4*)src, row_mask, relu_of, (float4*)out);
    else
      TEMP_LAUNCH(K_SEGMENT_SUM, k_segment_sum_rows_blk<4>, dim3(n_seg < 4096 ? n_seg : 4096), dim3(4 * 64), 0, st, n_seg, d4, seg_ptr, order,
                  (const float4*)src, row_mask, relu_of, (float4*)out);
    return launch_status();
  }
  const bool short_segs = n_rows_hint > 0 && n_rows_hint <= 2LL * n_seg;      // four segments per wave in lockstep
  int grid = ceil_div(n_seg, short_segs ? 16 : 4);
  if (grid > 2048) grid = 2048;
#define TEMP_SEGSUM(L)                                                                                                                      \
  do {                                                                                                                                      \
    if (short_segs) TEMP_LAUNCH(K_SEGMENT_SUM, (k_segment_sum_rows_short<L>), dim3(grid), dim3(256), 0, st, n_seg, d4, seg_ptr, order,      \
                                (const float4*)src, row_mask, relu_of, (float4*)out);                                                       \
    else TEMP_LAUNCH(K_SEGMENT_SUM, (k_segment_sum_rows<L>), dim3(grid), dim3(256), 0, st, n_seg, d4, seg_ptr, order, (const float4*)src,   \
                     row_mask, relu_of, (float4*)out);                                                                                      \
  } while (0)
  if (d4 <= 8) TEMP_SEGSUM(8); else if (d4 <= 16) TEMP_SEGSUM(16); else if (d4 <= 32) TEMP_SEGSUM(32); else TEMP_SEGSUM(64);
#undef TEMP_SEGSUM
  return launch_status();
}
}  // namespace temp
extern "C" {

size_t temp_segment_sum_rows_workspace(int n_seg, int n_rows, int d) { return (n_seg <= 0 || d <= 0) ? 0 : segment_sum_rows_workspace(n_seg, n_rows, d); }

int temp_segment_sum_rows(int n_seg, int n_rows, int d, const int32_t* seg_ptr, const int32_t* order, const float* src, float* out,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (n_seg < 0 || d <= 0 || (n_seg > 0 && (!seg_ptr || !src || !out))) return TEMP_E_BADARG;
  return segment_sum_rows(n_seg, d, seg_ptr, order, src, nullptr, out, (hipStream_t)stream, n_rows, workspace, workspace_bytes);
}

int temp_segment_sum_rows_relu(int n_seg, int n_rows, int d, const int32_t* seg_ptr, const int32_t* order, const float* src, const float* relu_of,
                               float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_seg < 0 || d <= 0 || (n_seg > 0 && (!seg_ptr || !src || !out || !relu_of))) return TEMP_E_BADARG;
  return segment_sum_rows(n_seg, d, seg_ptr, order, src, nullptr, out, (hipStream_t)stream, n_rows, workspace, workspace_bytes, relu_of);
}

struct EpiPlainStore {
  float* out; int ldo;
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row_ctx(int) const { return RowCtx(); }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int, int) const { return zero4(); }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4) const { st4(out + (size_t)row * ldo + col, acc); }
};

// Partial product of one k-slice (k_gemm_panel with K split over the blocks, gemm_panel.hpp): slice s of the launch writes
// part[s * slice_elems + row * ldo + col]; k_reduce_slices adds the slices in order.
struct EpiPartialStore {
  static constexpr int k_split_tag = 1;
  float* part; int ldo; size_t slice_elems; int kc, col_blocks;
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row_ctx(int) const { return RowCtx(); }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int, int) const { return zero4(); }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4) const {
    st4(part + (size_t)(blockIdx.y / col_blocks) * slice_elems + (size_t)row * ldo + col, acc);
  }
};

// C^T written: out[col * ldo + row].  A lane of the weights-resident kernel owns one row and four consecutive columns, so for a
// fixed column the 32 lanes of a half-wave write 32 consecutive floats: coalesced 128-B segments.
struct EpiPlainStoreT {
  static constexpr int has_n_valid = 1;          // (gemm_bxr.hpp EpiColLimit: the weights-resident kernel reads no row of B past n_valid)
  float* out; int ldo;
  int n_valid;                                   // columns this problem has (<= the launch's N; columns past it are not stored)
  struct RowCtx {};
  __device__ __forceinline__ RowCtx row_ctx(int) const { return RowCtx(); }
  __device__ __forceinline__ float4 pre4(const RowCtx&, int, int) const { return zero4(); }
  __device__ __forceinline__ void fin4(const RowCtx&, int row, int col, float4 acc, float4) const {
    float* o = out + (size_t)col * ldo + row;
    if (col + 3 < n_valid) { o[0] = acc.x; o[ldo] = acc.y; o[2 * (size_t)ldo] = acc.z; o[3 * (size_t)ldo] = acc.w; return; }
    if (col < n_valid) o[0] = acc.x;
    if (col + 1 < n_valid) o[ldo] = acc.y;
    if (col + 2 < n_valid) o[2 * (size_t)ldo] = acc.z;
  }
};

int temp_linear_t(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int trans_b, float* Ct, int ldct, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || !B || (M > 0 && (!A || !Ct)) || ldct < M) return TEMP_E_BADARG;
  return launch_gemm_panel(K_GEMM_LINEAR, M, N, K, A, lda, nullptr, B, ldb, trans_b, EpiPlainStoreT{Ct, ldct, N}, (hipStream_t)stream);
}

int temp_linear(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int trans_b, float* C, int ldc, void* stream) {
  return temp_linear_keys(M, N, K, A, lda, nullptr, B, ldb, trans_b, C, ldc, stream);
}

int temp_linear_keys(int M, int N, int K, const float* A, int lda, const uint32_t* a_keys, const float* B, int ldb, int trans_b, float* C, int ldc,
                     void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || !B || (M > 0 && (!A || !C))) return TEMP_E_BADARG;
  if (ldc % 4) return TEMP_E_UNSUPPORTED;
  if (M <= 0) return TEMP_OK;
  PanelBatch<EpiPlainStore> batch;
  for (int i = 0; i < PANEL_MAXP; ++i) batch.p[i] = PanelProblem<EpiPlainStore>{0, nullptr, nullptr, nullptr, EpiPlainStore{C, ldc}};
  batch.p[0] = PanelProblem<EpiPlainStore>{M, A, nullptr, B, EpiPlainStore{C, ldc}, a_keys};
  return launch_gemm_panel_multi(K_GEMM_LINEAR, batch, 1, N, K, lda, ldb, trans_b, (hipStream_t)stream);
}

int temp_absmax_keys(int n, int d, const float* x, int ldx, uint32_t* row_keys, uint32_t* col_keys, void* stream) {
  if (n < 0 || d <= 0 || (n > 0 && !x)) return TEMP_E_BADARG;
  if (d % 4 || d > 256 || ldx % 4) return TEMP_E_UNSUPPORTED;
  if (n == 0 || (!row_keys && !col_keys)) return TEMP_OK;
  launch_absmax_keys(n, d, x, ldx, row_keys, col_keys, col_keys ? col_keys + d : nullptr, (hipStream_t)stream);
  return launch_status();
}
size_t temp_keys_cols_size(int d) { return d > 0 ? (size_t)(1 + ABSMAX_BLOCKS) * d : 0; }

int temp_gather_rows_keys(int n, int d, const float* table, const int32_t* idx, float* out, uint32_t* row_keys, uint32_t* col_keys, void* stream) {
  if (n < 0 || d <= 0 || (n > 0 && (!table || !idx || !out))) return TEMP_E_BADARG;
  if (d % 4 || d > 256) return TEMP_E_UNSUPPORTED;
  if (n == 0) return TEMP_OK;
  launch_gather_rows_keys(n, d, table, idx, out, row_keys, col_keys, col_keys ? col_keys + d : nullptr, (hipStream_t)stream);
  return launch_status();
}

static int bilinear_query_args(int P, int d, int kind, const void* a, const void* b, const void* c, const void* e, const void* f) {
  if (P < 0 || d <= 0 || (kind != TEMP_SCORE_DISTMULT && kind != TEMP_SCORE_COMPLEX)) return TEMP_E_BADARG;
  if (kind == TEMP_SCORE_COMPLEX ? d % 8 : d % 4) return TEMP_E_UNSUPPORTED;
  if (P > 0 && (!a || !b || !c || !e || (kind == TEMP_SCORE_COMPLEX && !f))) return TEMP_E_BADARG;
  return TEMP_OK;
}

int temp_bilinear_query_fwd(int P, int d, int kind, const float* ent_rows, const int32_t* known_idx, const float* rel, const int32_t* rel_idx,
                            const int32_t* is_tail, float* q, void* stream) {
  int rc = bilinear_query_args(P, d, kind, ent_rows, known_idx, rel, rel_idx, is_tail);
  if (rc != TEMP_OK || P == 0) return rc;
  if (!q) return TEMP_E_BADARG;
  int grid = ceil_div((long long)P * (d / 4), 256);
  if (grid > 8192) grid = 8192;
  TEMP_LAUNCH(K_GATHER_CE, k_bilinear_query<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, d, kind, ent_rows, known_idx, rel, rel_idx, is_tail,
              (const float*)nullptr, q, (float*)nullptr);
  return launch_status();
}

int temp_bilinear_query_bwd(int P, int d, int kind, const float* ent_rows, const int32_t* known_idx, const float* rel, const int32_t* rel_idx,
                            const int32_t* is_tail, const float* d_q, float* d_known_rows, float* d_rel_rows, void* stream) {
  int rc = bilinear_query_args(P, d, kind, ent_rows, known_idx, rel, rel_idx, is_tail);
  if (rc != TEMP_OK || P == 0) return rc;
  if (!d_q || !d_known_rows || !d_rel_rows) return TEMP_E_BADARG;
  int grid = ceil_div((long long)P * (d / 4), 256);
  if (grid > 8192) grid = 8192;
  TEMP_LAUNCH(K_GATHER_CE, k_bilinear_query<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, d, kind, ent_rows, known_idx, rel, rel_idx, is_tail,
              d_q, d_known_rows, d_rel_rows);
  return launch_status();
}

int temp_linear_multi(int count, const TempLinearProblem* probs, int N, int K, int lda, int ldb, int trans_b, int ldc, void* stream) {
  if (count < 0 || N <= 0 || K <= 0 || (count > 0 && !probs)) return TEMP_E_BADARG;
  if (ldc % 4) return TEMP_E_UNSUPPORTED;
  for (int i = 0; i < count; ++i)
    if (probs[i].M < 0 || !probs[i].B || (probs[i].M > 0 && (!probs[i].A || !probs[i].C))) return TEMP_E_BADARG;
  for (int i0 = 0; i0 < count; i0 += PANEL_MAXP) {
    const int n = count - i0 < PANEL_MAXP ? count - i0 : PANEL_MAXP;
    // Few rows against a long K (d_q = d_scores . all_entities: ~200 rows x 10 000 entities x D per window): the row panels
    // x column tiles are a few dozen blocks, each walking all of K -- 330 us on an idle chip.  K is cut into slices, every
    // (panel, tile, slice) is a block, the slices' partial products land in a scratch slot of the library and are summed in order.
    if (!trans_b && K >= 4096 && N <= 256 && ldc == N && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0) {
      long long panels = 0, rows = 0;
      int max_m = 0;
      for (int i = 0; i < n; ++i) {
        const int M = probs[i0 + i].M;
        panels += ceil_div(M, 128); rows += M; max_m = M > max_m ? M : max_m;
      }
      const int ntiles = ceil_div(N, 32);
      int S = (int)(768 / (panels * ntiles > 0 ? panels * ntiles : 1));
      if (S > K / 320) S = K / 320;
      const long long cap = (long long)BX_SLOT_BYTES / ((rows > 0 ? rows : 1) * ldc * (long long)sizeof(float));
      if (S > cap) S = (int)cap;
      if (S >= 2) {
        const int kc = ceil_div(ceil_div(K, S), GEMM_KC) * GEMM_KC;
        S = ceil_div(K, kc);
        float* scratch = (float*)bx_scratch((hipStream_t)stream, (size_t)S * rows * ldc * sizeof(float));
        if (S >= 2 && scratch) {
          PanelBatch<EpiPartialStore> pb;
          float* part[PANEL_MAXP];
          // outputs that follow each other in memory (the row blocks of one matrix): slices of the WHOLE row range, one reduction
          bool adjacent = true;
          for (int i = 1; i < n; ++i) adjacent = adjacent && probs[i0 + i].C == probs[i0 + i - 1].C + (size_t)probs[i0 + i - 1].M * ldc;
          size_t off = 0;
          for (int i = 0; i < PANEL_MAXP; ++i) {
            const TempLinearProblem& q = probs[i0 + (i < n ? i : 0)];
            const int M = i < n ? q.M : 0;
            part[i] = scratch + off;
            pb.p[i] = PanelProblem<EpiPartialStore>{M, q.A, nullptr, q.B,
                                                    EpiPartialStore{part[i], ldc, adjacent ? (size_t)rows * ldc : (size_t)M * ldc, kc, ntiles}};
            off += adjacent ? (size_t)M * ldc : (size_t)S * M * ldc;
          }
          TEMP_LAUNCH(K_GEMM_LINEAR, (k_gemm_panel<1, EpiPartialStore>), dim3(ceil_div(max_m, 128), ntiles * S, n), dim3(256), 0,
                      (hipStream_t)stream, pb, N, K, lda, ldb, trans_b, 0);
          if (adjacent) {
            reduce_slices(S, (size_t)rows * ldc, ldc, scratch, probs[i0].C, ldc, (hipStream_t)stream);
          } else {
            for (int i = 0; i < n; ++i)
              if (probs[i0 + i].M > 0)
                reduce_slices(S, (size_t)probs[i0 + i].M * ldc, ldc, part[i], probs[i0 + i].C, ldc, (hipStream_t)stream);
          }
          if (launch_status() != TEMP_OK) return TEMP_E_LAUNCH;
          continue;
        }
      }
    }
    // Few rows against MANY columns (scores_b = q_b . all_b^T: a few hundred queries x 7 000 - 10 000 entities per window): the
    // row panels of such a problem are mostly padding and every block streams its own tile of B.  Taken as C^T = B . A^T the
    // entities are the ROWS (tens of thousands over the launch's windows) and the queries the resident weights of the split-operand
    // kernel (gemm_bxr.hpp), the result written transposed -- 57 -> 25 us per four-window launch at the S-icews0515 shape.  The
    // windows' query counts differ: the launch takes the widest, every problem carries its own limit (EpiPlainStoreT::n_valid).
    if (trans_b && N >= 2048 && bx_enabled() && K % 8 == 0 && lda % 4 == 0 && ldb % 4 == 0 && !(option(TEMP_OPT_DEBUG) & 0x40000)) {      // (TEMP_DEBUG bit 18: A/B, as stored)
      int max_m = 0, min_m = 1 << 30;
      for (int i = 0; i < n; ++i) { const int M = probs[i0 + i].M; max_m = M > max_m ? M : max_m; min_m = M < min_m ? M : min_m; }
      const int Nt = (max_m + 3) & ~3;
      BxGeom bg;
      int G;
      if (min_m > 0 && max_m <= 1024 && bx_plan(Nt, K, ldb, lda, 1, N, (long long)n * N, &bg, &G)) {
        PanelBatch<EpiPlainStoreT> tb;
        for (int i = 0; i < PANEL_MAXP; ++i) {
          const TempLinearProblem& q = probs[i0 + (i < n ? i : 0)];
          tb.p[i] = PanelProblem<EpiPlainStoreT>{i < n ? N : 0, q.B, nullptr, q.A, EpiPlainStoreT{q.C, ldc, q.M}};
        }
        if (launch_bxr(K_GEMM_LINEAR, tb, n, bg, (hipStream_t)stream, nullptr)) {
          if (launch_status() != TEMP_OK) return TEMP_E_LAUNCH;
          continue;
        }
      }
    }
    PanelBatch<EpiPlainStore> batch;
    for (int i = 0; i < PANEL_MAXP; ++i) {
      const TempLinearProblem& q = probs[i0 + (i < n ? i : 0)];
      batch.p[i] = PanelProblem<EpiPlainStore>{i < n ? q.M : 0, q.A, nullptr, q.B, EpiPlainStore{q.C, ldc}};
    }
    const int rc = launch_gemm_panel_multi(K_GEMM_LINEAR, batch, n, N, K, lda, ldb, trans_b, (hipStream_t)stream);
    if (rc != TEMP_OK) return rc;
  }
  return TEMP_OK;
}

size_t temp_linear_tn_workspace(int M, int Ka, int Nb) { return (M < 0 || Ka <= 0 || Nb <= 0) ? 0 : gemm_tn_workspace(M, Ka, Nb) + 256; }

int temp_linear_tn(int M, int Ka, int Nb, const float* A, int lda, const float* B, int ldb, float* out, int ldo, void* workspace,
                   size_t workspace_bytes, void* stream) {
  if (M < 0 || Ka <= 0 || Nb <= 0 || !out || (M > 0 && (!A || !B))) return TEMP_E_BADARG;
  if (!workspace || workspace_bytes < temp_linear_tn_workspace(M, Ka, Nb)) return TEMP_E_WORKSPACE;
  return gemm_tn(M, Ka, Nb, A, lda, B, ldb, out, ldo, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t temp_linear_tn_multi_workspace(int count, int max_m, int Ka, int Nb) {
  if (count <= 0 || max_m < 0 || Ka <= 0 || Nb <= 0) return 0;
  size_t w = gemm_tn_workspace(max_m, Ka, Nb) + 256;                 // the per-problem loop it falls back to
  for (int i0 = 0; i0 < count; i0 += TN_MAXP) {
    const size_t m = gemm_tn_multi_workspace(count - i0 < TN_MAXP ? count - i0 : TN_MAXP, max_m, Ka, Nb) + 256;
    if (m > w) w = m;
  }
  return w;
}

int temp_linear_tn_multi(int count, const TempLinearProblem* probs, int Ka, int Nb, int lda, int ldb, int ldc, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (count < 0 || Ka <= 0 || Nb <= 0 || (count > 0 && !probs)) return TEMP_E_BADARG;
  int max_m = 0;
  for (int i = 0; i < count; ++i) {
    if (probs[i].M < 0 || !probs[i].C || (probs[i].M > 0 && (!probs[i].A || !probs[i].B))) return TEMP_E_BADARG;
    if (probs[i].M > max_m) max_m = probs[i].M;
  }
  if (count == 0) return TEMP_OK;
  if (!workspace || workspace_bytes < temp_linear_tn_multi_workspace(count, max_m, Ka, Nb)) return TEMP_E_WORKSPACE;
  for (int i0 = 0; i0 < count; i0 += TN_MAXP) {
    const int n = count - i0 < TN_MAXP ? count - i0 : TN_MAXP;
    int Ms[TN_MAXP];
    const float *As[TN_MAXP], *Bs[TN_MAXP];
    float* Cs[TN_MAXP];
    for (int i = 0; i < n; ++i) { Ms[i] = probs[i0 + i].M; As[i] = probs[i0 + i].A; Bs[i] = probs[i0 + i].B; Cs[i] = probs[i0 + i].C; }
    int rc = gemm_tn_multi(n, Ms, Ka, Nb, As, lda, Bs, ldb, Cs, ldc, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc == TEMP_E_UNSUPPORTED) {                                  // shapes of the split-operand kernels etc.: one product at a time
      for (int i = 0; i < n; ++i) {
        rc = gemm_tn(Ms[i], Ka, Nb, As[i], lda, Bs[i], ldb, Cs[i], ldc, workspace, workspace_bytes, (hipStream_t)stream);
        if (rc != TEMP_OK) return rc;
      }
    } else if (rc != TEMP_OK) {
      return rc;
    }
  }
  return TEMP_OK;
}

int temp_gather_ce_fwd(int P, int C, int N, const float* scores, const int32_t* cand, float* loss_rows, float* lse_rows, void* stream) {
  if (P < 0 || C <= 0 || N <= 0 || (P > 0 && (!scores || !cand || !loss_rows || !lse_rows))) return TEMP_E_BADARG;
  if (P == 0) return TEMP_OK;
  if (2 * (long long)C >= N && N <= 1024)
    TEMP_LAUNCH(K_GATHER_CE, k_gather_ce_fwd_cnt_w, dim3(ceil_div(P, 4)), dim3(256), (size_t)4 * N * sizeof(int), (hipStream_t)stream, P, C, N, scores, cand, loss_rows, lse_rows);
  else if (2 * (long long)C >= N && (size_t)N * sizeof(int) <= 64 * 1024)
    TEMP_LAUNCH(K_GATHER_CE, k_gather_ce_fwd_cnt, dim3(P), dim3(256), (size_t)N * sizeof(int), (hipStream_t)stream, C, N, scores, cand, loss_rows, lse_rows);
  else
    TEMP_LAUNCH(K_GATHER_CE, k_gather_ce_fwd, dim3(P), dim3(256), 0, (hipStream_t)stream, C, N, scores, cand, loss_rows, lse_rows);
  return launch_status();
}

int temp_gather_ce_bwd(int P, int C, int N, const float* scores, const int32_t* cand, const float* lse_rows, const float* scale,
                       float inv_rows, const float* row_scale, float* d_scores, void* stream) {
  if (P < 0 || C <= 0 || N <= 0 || !scale || (P > 0 && (!scores || !cand || !lse_rows || !d_scores))) return TEMP_E_BADARG;
  if ((size_t)N * sizeof(float) > 160 * 1024 - 1024) return TEMP_E_UNSUPPORTED;
  if (P == 0) return TEMP_OK;
  if (N <= 1024) {
    TEMP_LAUNCH(K_GATHER_CE, k_gather_ce_bwd_w, dim3(ceil_div(P, 4)), dim3(256), (size_t)4 * N * sizeof(int), (hipStream_t)stream, P, C, N, scores, cand, lse_rows, scale,
                inv_rows, row_scale, d_scores);
    return launch_status();
  }
  const size_t lds = (size_t)N * sizeof(float);
  if (lds > 65536) {
    if (hipFuncSetAttribute((const void*)k_gather_ce_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TEMP_E_LAUNCH;
  }
  TEMP_LAUNCH(K_GATHER_CE, k_gather_ce_bwd, dim3(P), dim3(256), lds, (hipStream_t)stream, C, N, scores, cand, lse_rows, scale, inv_rows, row_scale, d_scores);
  return launch_status();
}

int temp_assemble_views(int n_pieces, const int32_t* piece_desc, const int32_t* piece_start, const TempCopyDesc* descs, const int32_t* table,
                        int32_t* out, void* stream) {
  if (n_pieces < 0 || (n_pieces > 0 && (!piece_desc || !piece_start || !descs || !out))) return TEMP_E_BADARG;
  if (n_pieces == 0) return TEMP_OK;
  TEMP_LAUNCH(K_COPY, k_assemble_views, dim3(n_pieces), dim3(256), 0, (hipStream_t)stream, piece_desc, piece_start, descs, table, out);
  return launch_status();
}

int temp_corrupt_sample(int R, int K, int N, uint64_t seed, const int32_t* truth, const int32_t* lo, const int32_t* hi, const int32_t* ids,
                        int32_t* cand, void* stream) {
  if (R < 0 || K < 0 || N <= 0 || (R > 0 && (!truth || !cand)) || ((lo != nullptr) != (hi != nullptr))) return TEMP_E_BADARG;
  if (R == 0) return TEMP_OK;
  const long long total = (long long)R * (K + 1);
  int grid = ceil_div(total, 256);
  if (grid > 16384) grid = 16384;
  TEMP_LAUNCH(K_GATHER_CE, k_corrupt_sample, dim3(grid), dim3(256), 0, (hipStream_t)stream, total, K + 1, N, (unsigned long long)seed, truth, lo, hi, ids, cand);
  return launch_status();
}

int temp_filtered_rank(int P, int N, int ld, const float* scores, const int32_t* target, const int32_t* filt_ptr,
                       const int32_t* filt_ids, int32_t* ranks, void* stream) {
  if (P < 0 || N <= 0 || ld < N || ld % 4 || (P > 0 && (!scores || !target || !ranks))) return TEMP_E_BADARG;
  if (P == 0) return TEMP_OK;
  TEMP_LAUNCH(K_GATHER_CE, k_filtered_rank, dim3(P), dim3(256), 0, (hipStream_t)stream, N, ld, scores, target, filt_ptr, filt_ids, ranks);
  return launch_status();
}

int temp_copy_probe(const void* src, void* dst, size_t bytes, void* stream) {
  if (!src || !dst || bytes % 16) return TEMP_E_BADARG;
  if (bytes == 0) return TEMP_OK;
  TEMP_LAUNCH(K_COPY, k_copy, dim3(2048), dim3(256), 0, (hipStream_t)stream, bytes / 16, (const float4*)src, (float4*)dst);
  return launch_status();
}

}  // extern "C"
