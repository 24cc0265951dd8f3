// Weights-resident fp32 MFMA GEMM:  C[M, N] = epi( A[M,K] . B )  for the skinny shapes of this
// path (K, N = D or 3D with D = 200; M = tens of thousands of node rows).
//
// B (a weight matrix, <= 480 KB) is cut into column slices of <= 3 MFMA tiles (96 columns); a
// block copies ITS slice into LDS once ([col][k], row stride K+4 or K+8 floats so that the
// ds_read_b128 of the 16-lane service groups hit 16 distinct bank quads) and then streams row
// panels through it: no barrier after the prologue, every wave walks its own 32-row panels
//   A fragment: one float4 per lane per 4 MFMAs straight from global memory (k = 8q + 4hh .. +3),
//   B fragment: one ds_read_b128 per lane per 4 MFMAs (the same four k),
//   MFMA operands swapped (weights as the A operand) so a lane ends up owning ONE output row and
//   4 consecutive columns per accumulator quad -> float4 epilogue, same Epi contract as gemm_panel.
// 80 KB of LDS per block -> 2 blocks (8 waves) per CU: one wave's global loads / epilogue overlap
// the other's MFMAs.  Blocks b, b+8, b+16 ... share an XCD (b % 8): each XCD walks a contiguous
// eighth of the row panels for ALL column slices, so the A panel a slice-block reads is in that
// XCD's L2 when the other slices' blocks ask for it.  Blocks per slice are proportional to its
// tile count.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "gemm_panel.hpp"
#include "gemm_bx.hpp"

namespace temp {

#define WRES_TPS 3                       // tiles (32 columns) per resident slice
#define WRES_LDS_BYTES (80 * 1024)
#define WRES_LOCAL_BLOCKS 64             // blocks per XCD (2 per CU)

#define WRES_MIN_ROWS 4096
#define WRES_QC 5                        // q-steps (8 k each) per software-pipeline stage: 40 k, 5 float4 of A per lane

struct WresGeom {
  int N, K, lda, ldb, trans_b;
  int ldk;                               // LDS row stride (floats)
  int kpad;                              // K rounded up to 40 (zero padded in LDS)
  int n_tiles, tps;                      // tps tiles per slice (<= n_tiles)
  int n_slices;                          // the last slice starts at tile n_tiles - tps and stores only its last `tail_store` tiles
  int tail_store;                        // (it overlaps its predecessor when tps does not divide n_tiles)
  int gate_stride;                       // 0: plain GEMM.  > 0 (= D): GRU mode -- slice j owns columns [32j, 32j+32) of each of the
                                         // 3 gate blocks of B ([3D][K] stored [n][k]); tile t = gate t at row t*D + 32j of B
  int split;                             // 1: every block serves ONE problem of the batch (roles = problems x slices)
};

// An epilogue that needs all tiles of a slice at once (the GRU cell: gates r, z, n of one column) declares
//   typedef ... GroupPre;  GroupPre pre_group(int row, bool row_ok, long a_src, int j0, int hh) const;   // issue its loads early
//     (a_src = the A row this output row was computed from: a_idx[row], row itself without a_idx, -1 for a zero row)
//   void fin_group(const GroupPre&, int row, bool row_ok, int j0, int hh, const f32x16 (&acc)[3]) const;
template <class Epi, class = void> struct EpiIsGroup { static constexpr bool value = false; };
template <class Epi> struct EpiIsGroup<Epi, std::void_t<typename Epi::GroupPre>> { static constexpr bool value = true; };
template <class Epi, bool G = EpiIsGroup<Epi>::value> struct EpiGroupPre { struct type {}; };
template <class Epi> struct EpiGroupPre<Epi, true> { typedef typename Epi::GroupPre type; };

// Host-side feasibility + geometry.  Returns false when the shape should use the streaming kernel.
inline bool wres_plan(int N, int K, int lda, int ldb, int trans_b, long long total_rows, WresGeom* g) {
  if (K % 4 || lda % 4 || ldb % 4 || N % 4 || K < 8) return false;
  g->N = N; g->K = K; g->lda = lda; g->ldb = ldb; g->trans_b = trans_b;
  g->gate_stride = 0; g->split = 0;
  g->kpad = (K + 8 * WRES_QC - 1) / (8 * WRES_QC) * (8 * WRES_QC);
  g->ldk = g->kpad + 4;                  // kpad / 4 is even, so ldk / 4 is odd: conflict-free ds_read_b128 (see header)
  g->n_tiles = ceil_div(N, 32);
  int tmax = WRES_TPS;
  while (tmax > 1 && (size_t)tmax * 32 * g->ldk * 4 > WRES_LDS_BYTES) --tmax;
  if ((size_t)tmax * 32 * g->ldk * 4 > WRES_LDS_BYTES) return false;
  if (tmax > g->n_tiles) tmax = g->n_tiles;
  // slices of `tps` tiles; when tps does not divide n_tiles the last slice is shifted left and recomputes
  // (without storing) tiles its predecessor owns.  Pick the width with the fewest tile computations,
  // charging a quarter tile per slice for the extra pass over A.
  int best = 1;
  float best_cost = 1e30f;
  for (int t = 1; t <= tmax; ++t) {
    const int ns = ceil_div(g->n_tiles, t);
    const float cost = ns * t + 0.25f * ns;
    if (cost <= best_cost) { best_cost = cost; best = t; }
  }
  g->tps = best;
  g->n_slices = ceil_div(g->n_tiles, best);
  g->tail_store = g->n_tiles - (g->n_slices - 1) * best;
  if (g->n_slices > 16) return false;                  // wide outputs (score matrix): stream B instead
  if (total_rows < WRES_MIN_ROWS) return false;        // tiny problems: the B prologue dominates
  return true;
}

// One launch covers all column slices (NTS tiles each); every XCD runs `bps` blocks per slice.
// VAR is 0 in the library; tools/wres_probe.hip instantiates ablations (bit0: no A loads, bit1: no epilogue,
// bit2: no LDS reads of B, bit3: per-wave s_memtime stamps through epi.stamp()).
template <int NTS, class Epi, int VAR = 0>
__global__ void __launch_bounds__(256, 2) k_gemm_wres(PanelBatch<Epi> batch, int count, WresGeom g, int bps) {
  extern __shared__ __attribute__((aligned(16))) float Ws[];
  constexpr int QC = WRES_QC;
  constexpr bool GROUP = EpiIsGroup<Epi>::value;
  static_assert(!GROUP || NTS == 3, "grouped epilogues own the three gate tiles of a slice");
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int role = local / bps, idx = local - role * bps;
  const int zsel = g.split ? role / g.n_slices : -1;          // split: this block serves problem zsel only
  const int slice = g.split ? role - zsel * g.n_slices : role;
  const bool tail = !GROUP && slice == g.n_slices - 1;
  const int n0 = GROUP ? slice * 32 : (tail ? g.n_tiles - NTS : slice * NTS) * 32;
  const int t_store = tail ? NTS - g.tail_store : 0;       // first tile of this slice that is stored
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int K = g.K, ldk = g.ldk;
  const int nch = g.kpad / (8 * QC);

  if constexpr (VAR & 8) batch.p[0].epi.stamp(0, __builtin_amdgcn_s_memtime());
  for (int z = 0; z < count; ++z) {
    const PanelProblem<Epi>& pb = batch.p[z];
    const int M = pb.M;
    if (M <= 0 || (zsel >= 0 && z != zsel)) continue;
    if (zsel >= 0 || z == 0 || pb.B != batch.p[z - 1].B) {
      if (zsel < 0 && z > 0) __syncthreads();
      // ---- prologue: this block's slice of B -> LDS as [col][k] (k contiguous), zero padded.
      // Loads are issued in batches of WRES_PRO per thread (a 96 x 200 slice = 19 float4 per thread: ONE batch, i.e. one
      // memory round trip) before the first LDS store of the batch.
      // No integer division here: with runtime divisors it cost ~13 k cycles per wave (measured), more than
      // the copy itself.  Threads are laid out (tx, ty) with power-of-two widths; 8 loads per thread in flight.
      const float* __restrict__ B = pb.B;
      constexpr int COLS = NTS * 32;
      const int k4n = g.kpad >> 2;
      auto col_of = [&](int c, bool* ok) {                       // LDS column c -> column of B
        const int n = GROUP ? (c >> 5) * g.gate_stride + n0 + (c & 31) : n0 + c;
        *ok = GROUP ? (n0 + (c & 31) < g.gate_stride) : (n < g.N);
        return n;
      };
      if (g.trans_b) {                                           // B stored [n][k]: lanes run along k
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
        for (int k4 = tx; k4 < k4n; k4 += 64) {
          const int k = k4 * 4;
          for (int c0 = ty; c0 < COLS; c0 += 32) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              bool okc;
              const int n = col_of(c0 + 4 * u, &okc);
              const bool ok = okc && k < K;
              v[u] = ld4(B + (ok ? (size_t)n * g.ldb + k : 0));
              if (!ok) v[u] = zero4();
            }
            if constexpr (VAR & 8) pb.epi.stamp(7, __builtin_amdgcn_s_memtime());
#pragma unroll
            for (int u = 0; u < 8; ++u) st4(Ws + (size_t)(c0 + 4 * u) * ldk + k, v[u]);
          }
        }
      } else {                                                   // B stored [k][n]: lanes run along n
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int c = tx; c < COLS; c += 32) {
          bool okc;
          const int n = col_of(c, &okc);
          for (int k40 = ty; k40 < k4n; k40 += 64) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int k = (k40 + 8 * u) * 4;
              const bool ok = okc && k < K;
              const float* b = B + (ok ? (size_t)k * g.ldb + n : 0);
              const size_t st = ok ? (size_t)g.ldb : 0;
              v[u] = make_float4(b[0], b[st], b[2 * st], b[3 * st]);
              if (!ok) v[u] = zero4();
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (k40 + 8 * u < k4n) st4(Ws + (size_t)c * ldk + (k40 + 8 * u) * 4, v[u]);
          }
        }
      }
      if constexpr (VAR & 8) pb.epi.stamp(5, __builtin_amdgcn_s_memtime());
      __syncthreads();
    }
    if constexpr (VAR & 8) pb.epi.stamp(1, __builtin_amdgcn_s_memtime());
    const float* __restrict__ A = pb.A;
    const int32_t* __restrict__ a_idx = pb.a_idx;
    const Epi& epi = pb.epi;
    const int panels = (M + 31) >> 5;
    const int per_xcd = (panels + 7) >> 3;
    const int p_end = min(panels, (xcd + 1) * per_xcd);
    const int stride = bps * 4;
    const float* wrow = Ws + (size_t)li * ldk + 4 * hh;

    // Software pipeline over the flat sequence of (panel, 40-k chunk) stages of this wave: the A
    // fragment of stage s+1 is in flight (5 float4 per lane) while the MFMAs of stage s run.
    int panel = xcd * per_xcd + idx * 4 + wave;
    int chunk = 0;
    int ld_panel = panel, ld_chunk = 0;
    const float* ld_ptr = A;
    bool ld_ok = false;
    long ld_src = -1;
    auto setup_ld = [&]() {
      long src = -1;
      const int r = ld_panel * 32 + li;
      if (ld_panel < p_end && r < M) src = a_idx ? (long)a_idx[r] : (long)r;
      ld_src = src;
      ld_ok = src >= 0;
      ld_ptr = A + (size_t)(ld_ok ? src : 0) * g.lda + 4 * hh;
    };
    auto issue = [&](float4 (&buf)[QC], bool& ok_out, int& kc_out, long& src_out) {
      ok_out = ld_ok;
      src_out = ld_src;
      kc_out = ld_chunk * 8 * QC;
#pragma unroll
      for (int q = 0; q < QC; ++q) {
        const bool ok = ld_ok && (kc_out + q * 8 + 4 * hh < K);
        if constexpr (VAR & 1) buf[q] = make_float4(1.f, 2.f, 3.f, 4.f);
        else buf[q] = ld4(ld_ptr + (ok ? kc_out + q * 8 : -4 * hh));  // !ok: k = 0..3 of a valid row, zeroed at use
      }
      if (++ld_chunk == nch) { ld_chunk = 0; ld_panel += stride; setup_ld(); }
    };
    f32x16 acc[NTS];
    auto zero_acc = [&]() {
#pragma unroll
      for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    };
    typename EpiGroupPre<Epi>::type gpre;
    auto stage = [&](float4 (&cur)[QC], bool cur_ok, int kc, long cur_src, float4 (&nxt)[QC], bool& nxt_ok, int& nxt_kc, long& nxt_src) {
      issue(nxt, nxt_ok, nxt_kc, nxt_src);          // unconditional (past the end: a harmless re-read of row 0) so the vmcnt waits stay exact
      if constexpr (GROUP) {               // the epilogue's own loads ride behind the MFMAs of the panel's last stage
        if (chunk == nch - 1) gpre = epi.pre_group(panel * 32 + li, panel * 32 + li < M, cur_src, n0, hh);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < QC; ++q) {
        const bool ok = cur_ok && (kc + q * 8 + 4 * hh < K);
        const float4 a = ok ? cur[q] : zero4();
        float4 w[NTS];
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
          if constexpr (VAR & 4) w[t] = make_float4(0.5f, 0.25f, 0.125f, 1.f);
          else w[t] = ld4(wrow + (size_t)t * 32 * ldk + kc + q * 8);
        }
#pragma unroll
        for (int t = 0; t < NTS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].x, a.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].y, a.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].z, a.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].w, a.w, acc[t], 0, 0, 0);
      }
      if (++chunk == nch) {
        // epilogue: lane (li, hh) owns output row; registers 4q..4q+3 of tile t are columns n0 + t*32 + 8q + 4hh .. +3
        const int row = panel * 32 + li;
        const bool row_ok = row < M;
        if constexpr (VAR & 8) epi.stamp(2, __builtin_amdgcn_s_memtime());
        if constexpr (GROUP) {
          epi.fin_group(gpre, row, row_ok, n0, hh, acc);
        } else {
        const typename Epi::RowCtx rc = epi.row_ctx(row_ok ? row : 0);
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
          if (t < t_store) continue;
          float4 pre[4];
          bool okc[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = n0 + t * 32 + 8 * q + 4 * hh;
            okc[q] = row_ok && col < g.N;
            pre[q] = epi.pre4(rc, okc[q] ? row : 0, okc[q] ? col : 0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = n0 + t * 32 + 8 * q + 4 * hh;
            if constexpr (VAR & 2) { if (acc[t][4 * q] == 12345.678f) epi.fin4(rc, row, col, zero4(), pre[q]); continue; }
            if (okc[q]) epi.fin4(rc, row, col, make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]), pre[q]);
          }
        }
        }
        if constexpr (VAR & 8) epi.stamp(3, __builtin_amdgcn_s_memtime());
        zero_acc();
        chunk = 0;
        panel += stride;
      }
    };
    if (panel < p_end) {
      float4 bufA[QC], bufB[QC];
      bool okA = false, okB = false;
      int kcA = 0, kcB = 0;
      long srcA = -1, srcB = -1;
      setup_ld();
      zero_acc();
      issue(bufA, okA, kcA, srcA);
      while (true) {
        stage(bufA, okA, kcA, srcA, bufB, okB, kcB, srcB);
        if (panel >= p_end) break;
        stage(bufB, okB, kcB, srcB, bufA, okA, kcA, srcA);
        if (panel >= p_end) break;
      }
    }
    if constexpr (VAR & 8) epi.stamp(4, __builtin_amdgcn_s_memtime());
  }
}

template <int NTS, class Epi>
int launch_wres_one(int kid, const PanelBatch<Epi>& batch, int count, const WresGeom& g, hipStream_t st, int local_blocks = WRES_LOCAL_BLOCKS) {
  static bool attr_set = false;          // per instantiation; idempotent
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_gemm_wres<NTS, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, WRES_LDS_BYTES) != hipSuccess)
      return TEMP_E_LAUNCH;
    attr_set = true;
  }
  const int roles = g.n_slices * (g.split ? count : 1);
  const size_t lds = (size_t)NTS * 32 * g.ldk * 4;
  // narrow slices leave LDS for more than two blocks per CU: up to 3 (12 waves) hide the A-stream latency better
  if (local_blocks == WRES_LOCAL_BLOCKS) {
    int per_cu = (int)((160 * 1024) / (lds + 1024));
    if (per_cu > 3) per_cu = 3;
    if (per_cu > 2) local_blocks = 32 * per_cu;
  }
  int bps = local_blocks / roles;
  if (bps < 1) bps = 1;
  // small problems (a window position at ICEWS scale is a few hundred rows): no more blocks than there are 32-row panels
  // to walk -- every extra block would still copy its 78 KB slice of B into LDS and find nothing to do
  int max_m = 0;
  for (int i = 0; i < count; ++i) max_m = batch.p[i].M > max_m ? batch.p[i].M : max_m;
  const int need = ceil_div(ceil_div(ceil_div(max_m, 32), 8), 4);
  if (bps > need) bps = need < 1 ? 1 : need;
  TEMP_LAUNCH(kid, (k_gemm_wres<NTS, Epi>), dim3(roles * bps * 8), dim3(256), lds, st, batch, count, g, bps);
  return launch_status();
}

template <class Epi>
int launch_gemm_wres(int kid, const PanelBatch<Epi>& batch, int count, const WresGeom& g, hipStream_t st) {
  if (g.tps == 3) return launch_wres_one<3, Epi>(kid, batch, count, g, st);
  if (g.tps == 2) return launch_wres_one<2, Epi>(kid, batch, count, g, st);
  return launch_wres_one<1, Epi>(kid, batch, count, g, st);
}

// Dispatcher used by every call site: weights-resident kernel when the shape allows, else the
// streaming row-panel kernel.  temp_set_option(TEMP_OPT_GEMM_STREAM, 1) forces the latter (A/B runs).
inline bool wres_disabled() { return option(TEMP_OPT_GEMM_STREAM) != 0; }

template <class Epi>
int launch_gemm_panel_multi(int kid, const PanelBatch<Epi>& batch, int count, int N, int K, int lda, int ldb, int trans_b, hipStream_t st) {
  if (count <= 0 || count > PANEL_MAXP) return TEMP_E_BADARG;
  long long rows = 0;
  for (int i = 0; i < count; ++i) rows += batch.p[i].M > 0 ? batch.p[i].M : 0;
  if (rows <= 0 || N <= 0) return TEMP_OK;
  if constexpr (!EpiIsGroup<Epi>::value) {
    // large products run on the bf16 matrix pipe with the exact three-way operand split (gemm_bx.hpp)
    if (bx_enabled()) {
      int max_m = 0;
      for (int i = 0; i < count; ++i) max_m = batch.p[i].M > max_m ? batch.p[i].M : max_m;
      BxGeom bg;
      int G;
      if (bx_plan(N, K, lda, ldb, trans_b, max_m, rows, &bg, &G)) return launch_gemm_bx(kid, batch, count, bg, G, st);
      // K = 4 (mod 8) -- e.g. a product over 500 entities: the slab-staged f16 kernel reads A in quads, the bf16 kernels do not
      if (K % 8 == 4 && bx_plan(N, K, lda, ldb, trans_b, max_m, rows, &bg, &G, true) && hx_supported(batch, count, bg)) {
        const int rc = launch_gemm_hx(kid, batch, count, bg, G, st);
        if (rc != TEMP_E_UNSUPPORTED) return rc;
      }
    }
  }
  WresGeom g;
  if (!wres_disabled() && wres_plan(N, K, lda, ldb, trans_b, rows, &g)) {
    // problems with different B matrices: give every problem its own blocks instead of re-staging B per problem
    bool distinct = count > 1;
    for (int i = 1; i < count; ++i) distinct = distinct && batch.p[i].B != batch.p[i - 1].B;
    if (distinct && g.n_slices * count <= WRES_LOCAL_BLOCKS) g.split = 1;
    return launch_gemm_wres(kid, batch, count, g, st);
  }
  return launch_gemm_stream_multi(kid, batch, count, N, K, lda, ldb, trans_b, st);
}

template <class Epi>
int launch_gemm_panel(int kid, int M, int N, int K, const float* A, int lda, const int32_t* a_idx, const float* B, int ldb, int trans_b,
                      const Epi& epi, hipStream_t st) {
  if (M <= 0 || N <= 0) return TEMP_OK;
  PanelBatch<Epi> batch;
  for (int i = 0; i < PANEL_MAXP; ++i) batch.p[i] = PanelProblem<Epi>{0, nullptr, nullptr, nullptr, epi};
  batch.p[0] = PanelProblem<Epi>{M, A, a_idx, B, epi};
  return launch_gemm_panel_multi(kid, batch, 1, N, K, lda, ldb, trans_b, st);
}

}  // namespace temp
