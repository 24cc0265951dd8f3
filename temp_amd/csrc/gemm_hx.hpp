// fp32 GEMM on the f16 matrix pipe:  C[M, N] = epi( A[M, K] . B )  as THREE MFMA products of the scaled two-way operand split
// (split_f16.hpp) -- the kernels of gemm_bx.hpp / gemm_bxr.hpp (six bf16 products of the exact three-way split) with half the MFMA
// count and two operand planes instead of three.
//
//   A (activations): every ROW carries a power-of-two scale from its key (the row's largest magnitude, PanelProblem::a_keys).  The
//      key comes from the producer of A where one exists (the chain backward writes the row keys of g4, gru_chain_hx.hpp); otherwise
//      the launcher takes it with one pass over A (k_absmax_keys: the rows are then in L2 / the Infinity Cache for the product).
//   B (weights): per-COLUMN keys; packed once per launch into two f16 planes in fragment order (hx_pack.hpp) or -- weights-resident
//      kernel -- cut by every workgroup for its own column tiles.
//   C: accumulated in fp32 on the scaled operands, unscaled by 1 / (row scale . column scale) in the epilogue (exact: powers of two).
#pragma once
#include "gemm_bx.hpp"
#include "hx_pack.hpp"

namespace temp {

struct HxPacked { const hx_u32x4* b[PANEL_MAXP]; const unsigned* keys[PANEL_MAXP]; };   // packed planes and column keys of every problem's B

// One row tile of 128 rows x G column tiles per block, the slab of B staged from the packed planes (k_gemm_bxp's structure).
// keys_by_out: a_keys is indexed by the OUTPUT row (the launcher's own pass over a gathered A); else by the source row a_idx[row].
template <int G, class Epi>
__global__ void __launch_bounds__(BX_THREADS, (G <= 4 ? 3 : 2)) k_gemm_hxp(PanelBatch<Epi> batch, BxGeom g, HxPacked packed, int keys_by_out) {
  constexpr int PIECES = G * 128;                             // 16-byte pieces of a slab of the group
  constexpr int NPC = (PIECES + BX_THREADS - 1) / BX_THREADS;
  __shared__ __attribute__((aligned(16))) hx_u32x4 Bs[2][PIECES];
  const PanelProblem<Epi>& pb = batch.p[blockIdx.y];
  const int M = pb.M;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int grp = local % g.n_groups, rt_local = local / g.n_groups;
  const int rt = xcd * g.per_xcd + rt_local;
  if (rt_local >= g.per_xcd || rt * 128 >= M) return;        // uniform
  const bool tail = grp == g.n_groups - 1;
  const int t0 = tail ? g.n_tiles - G : grp * G;
  const int t_store = tail ? G - g.tail_store : 0;
  const int n0 = t0 * 32;
  const int N = g.N, K = g.K;
  const float* __restrict__ A = pb.A;
  const int32_t* __restrict__ a_idx = pb.a_idx;
  const Epi& epi = pb.epi;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int row = rt * 128 + wave * 32 + li;
  long a_src = -1;
  if (row < M) a_src = a_idx ? (long)a_idx[row] : (long)row;
  // rows past M and gathered "zero rows" (a_idx < 0) compute on row 0; the former are never stored, the latter are zeroed
  // before the epilogue.  k past K meets the zero padding of the packed B.
  const float* aptr = A + (size_t)(a_src >= 0 ? a_src : 0) * g.lda;
  const unsigned akey = pb.a_keys[a_src >= 0 ? (keys_by_out ? (long)row : a_src) : 0];
  const float sa = hx_scale(akey);
  const int kclamp = K - 4;                                   // last quad that may be read (K % 4 == 0; a quad past it meets zeros of B)

  f32x16 acc[G];
#pragma unroll
  for (int t = 0; t < G; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const hx_u32x4* bsrc = packed.b[blockIdx.y] + (size_t)t0 * 128;
  const size_t slab_stride = (size_t)g.n_tiles * 128;
  int piece[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) piece[i] = (threadIdx.x + i * BX_THREADS) % PIECES;   // the overhang redoes early pieces
  hx_u32x4 br[NPC];
  const int nslabs = (K + 15) >> 4;
  auto fetch_b = [&](int s) {
    const hx_u32x4* p = bsrc + (size_t)(s < nslabs ? s : nslabs - 1) * slab_stride;
#pragma unroll
    for (int i = 0; i < NPC; ++i) br[i] = p[piece[i]];
  };
  auto fetch_a = [&](float4 (&a)[2], int k0) {
    const int k = k0 + 8 * hh;
    a[0] = ld4(aptr + (k <= kclamp ? k : kclamp));
    a[1] = ld4(aptr + (k + 4 <= kclamp ? k + 4 : kclamp));
  };
  float4 a1[2], a2[2];                                       // A of slabs s+1, s+2
  hx_u32x4 AH, AL, NH, NL;                                   // split A of slabs s, s+1
  // the non-MFMA work of a slab in chunks that fit an MFMA shadow: chunks 0..3 = element pairs of the next A fragment (6 VALU
  // instructions each), chunks 4.. = one LDS store of the next B slab each
  constexpr int NCHUNK = 4 + NPC;
  auto chunk = [&](int c, int buf) {
    if (c < 4) {
      const float4 f = a1[c >> 1];
      unsigned h, l;
      hx_split_pair((c & 1) ? f.z : f.x, (c & 1) ? f.w : f.y, sa, h, l);
      NH[c] = h; NL[c] = l;
    } else {
      Bs[buf][piece[c - 4]] = br[c - 4];
    }
  };

  fetch_b(0);
  fetch_a(a1, 0);
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) chunk(c, 0);
  AH = NH; AL = NL;
  fetch_a(a1, 16);
  __syncthreads();
  for (int s = 0; s < nslabs; ++s) {
    fetch_b(s + 1);                                          // unconditional (past the end: the last slab again)
    fetch_a(a2, s * 16 + 32);
    const hx_f16x8 ah = hx_frag(AH), al = hx_frag(AL);
    const hx_u32x4* bs = &Bs[s & 1][lane];
    // Tiles go through the matrix pipe in PAIRS (the three products of tile t alternate with those of tile t + 1: an MFMA never
    // waits for the accumulator of the one just issued); the fragments of the next pair are read from LDS behind the first MFMAs
    // of this pair; every second MFMA is followed by one chunk of the other work, the LDS stores of the next slab last.
    constexpr int NP = (G + 1) / 2;
    constexpr int NSLOT = (G * 3) / 2;                       // chunk slots behind the MFMAs
    constexpr int FIRST_B = NSLOT - NPC > 4 ? NSLOT - NPC : 4;   // slot of the first B store
    hx_u32x4 wf[2][2][2];                                    // [pair parity][tile of the pair][plane h, l]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if (u < G) wf[0][u][p] = bs[(u * 2 + p) * 64];
    __builtin_amdgcn_sched_barrier(0);
    int slot = 0;
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const bool two = 2 * pr + 1 < G;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !two) continue;
          const int t = 2 * pr + u;
          const hx_f16x8 wh = hx_frag(wf[pr & 1][u][0]), wl = hx_frag(wf[pr & 1][u][1]);
          // small terms first
          if (j == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah, acc[t], 0, 0, 0);
          if (j == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, acc[t], 0, 0, 0);
          if (j == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, acc[t], 0, 0, 0);
          if (j == 0 && pr + 1 < NP) {                       // the next pair's fragments
            const int tn = 2 * (pr + 1) + u;
            if (tn < G) {
#pragma unroll
              for (int p = 0; p < 2; ++p) wf[(pr + 1) & 1][u][p] = bs[(tn * 2 + p) * 64];
            }
          }
          if (slot & 1) {
            const int c = slot >> 1;
            if (c < 4) chunk(c, (s + 1) & 1);
            else if (c >= FIRST_B && c - FIRST_B + 4 < NCHUNK) chunk(c - FIRST_B + 4, (s + 1) & 1);
          }
          ++slot;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {                       // narrow groups: what found no slot behind an MFMA
      const bool done = c < 4 ? c < NSLOT : (FIRST_B + c - 4 < NSLOT);
      if (!done) chunk(c, (s + 1) & 1);
    }
    a1[0] = a2[0]; a1[1] = a2[1];
    AH = NH; AL = NL;
    __syncthreads();
  }

  const bool row_ok = row < M;
  if (a_idx && a_src < 0) {                                   // a gathered zero row
#pragma unroll
    for (int t = 0; t < G; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  }
  const typename Epi::RowCtx rc = epi.row_ctx(row_ok ? row : 0);
  const float ia = hx_inv_scale(akey);
  const unsigned* ckeys = packed.keys[blockIdx.y] + n0;
  // Epilogue in groups of up to four tiles: all the epilogue's own loads of a group are issued before its first store (gemm_bx.hpp)
  constexpr int EG = 4;
#pragma unroll
  for (int t0e = 0; t0e < G; t0e += EG) {
    float4 pre[EG][4];
    bool ok[EG][4];
#pragma unroll
    for (int u = 0; u < EG; ++u) {
      const int t = t0e + u;
      if (t >= G) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + t * 32 + 8 * q + 4 * hh;
        ok[u][q] = row_ok && col < N && t >= t_store;
        pre[u][q] = epi.pre4(rc, ok[u][q] ? row : 0, ok[u][q] ? col : 0);
      }
    }
#pragma unroll
    for (int u = 0; u < EG; ++u) {
      const int t = t0e + u;
      if (t >= G) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cl = t * 32 + 8 * q + 4 * hh, col = n0 + cl;
        if (ok[u][q]) {
          const float i0 = ia * hx_inv_scale(ckeys[cl]), i1 = ia * hx_inv_scale(ckeys[cl + 1]), i2 = ia * hx_inv_scale(ckeys[cl + 2]), i3 = ia * hx_inv_scale(ckeys[cl + 3]);
          epi.fin4(rc, row, col, make_float4(acc[t][4 * q] * i0, acc[t][4 * q + 1] * i1, acc[t][4 * q + 2] * i2, acc[t][4 * q + 3] * i3), pre[u][q]);
        }
      }
    }
  }
}

// ---- launch: scratch slot = [packed planes + column keys of the distinct weight matrices | row keys of the problems that bring none]
#define HX_KEYS_OFFSET (4u << 20)                             // row keys start here inside the slot (the packs use up to BX_PACK_MAX_BYTES)

inline size_t hx_pack_bytes(int N, int K) { return hx_packed_items(N, K) * 16 + (size_t)ceil_div(N, 32) * 32 * 4; }

// a_keys of a gathered / plain A by the launcher's own pass: keys per OUTPUT row
static __global__ void __launch_bounds__(256) k_absmax_rows_idx(int M, int K, const float* __restrict__ A, int lda, const int32_t* __restrict__ a_idx,
                                                                unsigned* __restrict__ keys) {
  const int lane = threadIdx.x & 63, k4 = K >> 2;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (int r0 = gw; r0 < M; r0 += 4 * nw) {
    float4 v[4];
    int src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int r = r0 + u * nw; src[u] = r < M ? (a_idx ? a_idx[r] : r) : -1; }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (src[u] >= 0 && lane < k4) ? ld4(A + (size_t)src[u] * lda + 4 * lane) : zero4();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * nw;
      unsigned k = hx_abs_bits4(v[u]);
      if (r < M) {                                            // (wave-uniform)
        for (int c = 256 + 4 * lane; c < K; c += 256) k = max(k, src[u] >= 0 ? hx_abs_bits4(ld4(A + (size_t)src[u] * lda + c)) : 0u);   // rows wider than a wave
        k = hx_wave_max(k);
        if (lane == 0) keys[r] = k;
      }
    }
  }
}

// the same for up to PANEL_MAXP problems of one launch (blockIdx.y = problem): a multi-problem product pays ONE key pass
struct AbsRowsBatch { int M[PANEL_MAXP]; const float* A[PANEL_MAXP]; const int32_t* idx[PANEL_MAXP]; unsigned* keys[PANEL_MAXP]; };
static __global__ void __launch_bounds__(256) k_absmax_rows_idx_multi(AbsRowsBatch b, int K, int lda) {
  const int pr = blockIdx.y;
  const int M = b.M[pr];
  const float* __restrict__ A = b.A[pr];
  const int32_t* __restrict__ a_idx = b.idx[pr];
  unsigned* __restrict__ keys = b.keys[pr];
  const int lane = threadIdx.x & 63, k4 = K >> 2;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (int r0 = gw; r0 < M; r0 += 4 * nw) {
    float4 v[4];
    int src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int r = r0 + u * nw; src[u] = r < M ? (a_idx ? a_idx[r] : r) : -1; }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (src[u] >= 0 && lane < k4) ? ld4(A + (size_t)src[u] * lda + 4 * lane) : zero4();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * nw;
      unsigned k = hx_abs_bits4(v[u]);
      if (r < M) {                                            // (wave-uniform)
        for (int c = 256 + 4 * lane; c < K; c += 256) k = max(k, src[u] >= 0 ? hx_abs_bits4(ld4(A + (size_t)src[u] * lda + c)) : 0u);
        k = hx_wave_max(k);
        if (lane == 0) keys[r] = k;
      }
    }
  }
}

template <class Epi>
static inline bool hx_supported(const PanelBatch<Epi>& batch, int count, const BxGeom& g) {
  if (!hx_enabled() || g.K % 4 || g.lda % 4) return false;
  size_t need_keys = 0;
  for (int i = 0; i < count; ++i)
    if (!batch.p[i].a_keys) need_keys += align_up((size_t)(batch.p[i].M > 0 ? batch.p[i].M : 0) * 4, 256);
  // A pass over A for its row keys costs ~13 us per 50 MB; the f16 kernels save ~0.2 us per (column tile x slab) of a 60 000-row
  // product.  Without caller keys the pass only pays for wide or deep products (measured: the 200 x 200 self-loop products lose
  // 10 us, the 200 x 600 gates gain 10, K = 600 gains 35).
  if (need_keys > 0 && ceil_div(g.N, 32) < 16 && g.K < 400 && !(option(TEMP_OPT_DEBUG) & 0x800)) return false;      // (TEMP_DEBUG bit 11: development A/B, take the pass anyway)
  return need_keys <= BX_SLOT_BYTES - HX_KEYS_OFFSET;
}

// Row keys of the problems that bring none: one pass over their rows into the slot's key region (keys per OUTPUT row).  All
// problems of a batch must then agree on the indexing: a batch that mixes caller keys (by source row) of gathered operands with
// keyless problems is refused (-> false).
template <class Epi>
static inline bool hx_fill_keys(PanelBatch<Epi>& batch, int count, int K, int lda, unsigned char* slot, hipStream_t st, int* keys_by_out) {
  bool any_own = false, any_given_gather = false;
  for (int i = 0; i < count; ++i) {
    if (batch.p[i].M <= 0) continue;
    if (!batch.p[i].a_keys) any_own = true;
    else if (batch.p[i].a_idx) any_given_gather = true;
  }
  if (any_own && any_given_gather) return false;
  size_t koff = HX_KEYS_OFFSET;
  AbsRowsBatch ab = {};
  int n_own = 0, max_own = 0, last = -1;
  for (int i = 0; i < count; ++i) {
    if (batch.p[i].M <= 0) { batch.p[i].a_keys = reinterpret_cast<const unsigned*>(slot + HX_KEYS_OFFSET); continue; }
    if (batch.p[i].a_keys) continue;
    unsigned* keys = reinterpret_cast<unsigned*>(slot + koff);
    koff += align_up((size_t)batch.p[i].M * 4, 256);
    ab.M[n_own] = batch.p[i].M; ab.A[n_own] = batch.p[i].A; ab.idx[n_own] = batch.p[i].a_idx; ab.keys[n_own] = keys;
    max_own = batch.p[i].M > max_own ? batch.p[i].M : max_own;
    ++n_own;
    last = i;
    batch.p[i].a_keys = keys;
  }
  if (n_own == 1) {
    int blocks = ceil_div(batch.p[last].M, 16);
    if (blocks > 2048) blocks = 2048;
    TEMP_LAUNCH(K_KEYS, k_absmax_rows_idx, dim3(blocks), dim3(256), 0, st, ab.M[0], K, ab.A[0], lda, ab.idx[0], ab.keys[0]);
  } else if (n_own > 1) {
    int blocks = ceil_div(max_own, 16);
    if (blocks > 2048 / n_own) blocks = 2048 / n_own;
    TEMP_LAUNCH(K_KEYS, k_absmax_rows_idx_multi, dim3(blocks, n_own), dim3(256), 0, st, ab, K, lda);
  }
  for (int i = count; i < PANEL_MAXP; ++i) batch.p[i].a_keys = batch.p[0].a_keys;
  *keys_by_out = any_own ? 1 : 0;
  return true;
}

template <int G, class Epi>
static inline void launch_hxp_g(int kid, const PanelBatch<Epi>& batch, int count, const BxGeom& g, hipStream_t st, const HxPacked& pk, int keys_by_out) {
  dim3 grid(8 * g.per_xcd * g.n_groups, count);
  TEMP_LAUNCH(kid, (k_gemm_hxp<G, Epi>), grid, dim3(BX_THREADS), 0, st, batch, g, pk, keys_by_out);
}

// -> TEMP_E_UNSUPPORTED when no scratch slot is free or the packs do not fit (the caller then takes the bf16 kernels)
template <class Epi>
int launch_gemm_hx(int kid, const PanelBatch<Epi>& batch_in, int count, const BxGeom& g, int G, hipStream_t st) {
  const size_t pbytes = align_up(hx_pack_bytes(g.N, g.K), 256);
  int n_distinct = 0, which[PANEL_MAXP];
  for (int i = 0; i < count; ++i) {
    which[i] = -1;
    for (int j = 0; j < i; ++j)
      if (batch_in.p[j].B == batch_in.p[i].B) { which[i] = which[j]; break; }
    if (which[i] < 0) which[i] = n_distinct++;
  }
  if (pbytes * n_distinct > BX_PACK_MAX_BYTES || n_distinct > HX_PACK_JOBS) return TEMP_E_UNSUPPORTED;
  unsigned char* slot = reinterpret_cast<unsigned char*>(bx_scratch(st, BX_SLOT_BYTES));
  if (!slot) return TEMP_E_UNSUPPORTED;
  PanelBatch<Epi> batch = batch_in;
  HxPacked pk;
  HxPackJobs jobs = {};
  const size_t items = hx_packed_items(g.N, g.K);
  int done = 0;
  for (int i = 0; i < PANEL_MAXP; ++i) { pk.b[i] = reinterpret_cast<const hx_u32x4*>(slot); pk.keys[i] = reinterpret_cast<const unsigned*>(slot + items * 16); }
  for (int i = 0; i < count; ++i) {
    unsigned char* dst = slot + (size_t)which[i] * pbytes;
    pk.b[i] = reinterpret_cast<const hx_u32x4*>(dst);
    pk.keys[i] = reinterpret_cast<const unsigned*>(dst + items * 16);
    if (which[i] < done) continue;
    ++done;
    hx_pack_jobs_add(jobs, batch.p[i].B, reinterpret_cast<hx_u32x4*>(dst), reinterpret_cast<unsigned*>(dst + items * 16), g.K, g.N, g.ldb, g.trans_b);
  }
  int kbo = 0;
  if (!hx_fill_keys(batch, count, g.K, g.lda, slot, st, &kbo)) return TEMP_E_UNSUPPORTED;
  hx_pack_launch(jobs, K_BX_PACK, st);
  hx_count();
  switch (G) {
    case 1: launch_hxp_g<1, Epi>(kid, batch, count, g, st, pk, kbo); break;
    case 2: launch_hxp_g<2, Epi>(kid, batch, count, g, st, pk, kbo); break;
    case 3: launch_hxp_g<3, Epi>(kid, batch, count, g, st, pk, kbo); break;
    case 4: launch_hxp_g<4, Epi>(kid, batch, count, g, st, pk, kbo); break;
    case 5: launch_hxp_g<5, Epi>(kid, batch, count, g, st, pk, kbo); break;
    case 6: launch_hxp_g<6, Epi>(kid, batch, count, g, st, pk, kbo); break;
    default: launch_hxp_g<7, Epi>(kid, batch, count, g, st, pk, kbo); break;
  }
  return launch_status();
}

}  // namespace temp
#include "gemm_hxr.hpp"
namespace temp {

// weights-resident f16 kernel (gemm_hxr.hpp) for short K -> false: not taken (the caller goes on to the bf16 resident kernel)
template <class Epi>
static inline bool launch_hxr(int kid, const PanelBatch<Epi>& batch_in, int count, const BxGeom& g, hipStream_t st) {
  if (!option(TEMP_OPT_GEMM_RESIDENT) || g.K > BXR_MAX_SLABS * 16 || g.K < 72 || g.K % 8 || !hx_supported(batch_in, count, g)) return false;
  int max_m = 0;
  for (int i = 0; i < count; ++i) max_m = batch_in.p[i].M > max_m ? batch_in.p[i].M : max_m;
  BxrGeom rg;
  if (!bxr_plan(g.N, g.K, g.lda, max_m, &rg)) return false;
  rg.ldb = g.ldb; rg.trans_b = g.trans_b;
  static const bool granted = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_hxr<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, HXR_LDS_BYTES) == hipSuccess;
  if (!granted) { (void)hipGetLastError(); return false; }
  PanelBatch<Epi> batch = batch_in;
  bool need = false;
  for (int i = 0; i < count; ++i) need = need || (batch.p[i].M > 0 && !batch.p[i].a_keys);
  int kbo = 0;
  if (need) {
    unsigned char* slot = reinterpret_cast<unsigned char*>(bx_scratch(st, BX_SLOT_BYTES));
    if (!slot || !hx_fill_keys(batch, count, g.K, g.lda, slot, st, &kbo)) return false;
  } else {
    for (int i = 0; i < PANEL_MAXP; ++i) if (!batch.p[i].a_keys) batch.p[i].a_keys = batch.p[0].a_keys;
    for (int i = 0; i < count; ++i) if (batch.p[i].M > 0) { for (int j = 0; j < PANEL_MAXP; ++j) if (!batch.p[j].a_keys) batch.p[j].a_keys = batch.p[i].a_keys; break; }
  }
  const size_t lds = (size_t)rg.n_slabs * BXR_G * 128 * 16 + 3 * BXR_BIAS_BYTES;
  TEMP_LAUNCH(kid, (k_gemm_hxr<Epi>), dim3(256, count), dim3(BXR_WAVES * 64), lds, st, batch, rg, kbo);
  hx_count();
  return true;
}

}  // namespace temp
