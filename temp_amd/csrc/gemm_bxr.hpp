// fp32 GEMM on the bf16 matrix pipe (exact three-way operand split, gemm_bx.hpp) for SHORT K: the packed weights stay RESIDENT
// in LDS.
//
//   C[M, N] = epi( A[M, K] . B ),   K <= 208 (13 slabs of 16: every K = 200 product of the step -- the GRU input gates, the
//                                             self-loop products and their transposes), M in the tens of thousands.
//
// k_gemm_bxp stages one 16-k slab of B per block and iteration (a barrier per slab) and gives every block ONE row tile: with 13
// slabs a block is mostly prologue and epilogue, all blocks of the launch run those phases at the same time (the chip alternates
// between a load burst, a compute phase and a store burst), and 641 row tiles on 512 resident blocks leave a second round that is
// a quarter full -- measured 0.15-0.3 of the pipe.  Here a workgroup (one per CU, 8 waves) copies the three bf16 planes of FOUR
// column tiles for ALL of K into LDS once (4 x 13 x 3 KB = 156 of the CU's 160 KB) and then every wave streams 32-row panels of A
// through them on its own: no barrier after the staging, no B traffic, A prefetched three slabs ahead ACROSS panel boundaries, the
// epilogue's own loads (the addend of the self-loop layer) issued behind the last slabs of the panel -- the waves of the chip
// drift apart, so loads, MFMAs and stores of different waves overlap.  The column tiles beyond four go to other CUs: the 32 block
// slots of an XCD are dealt to the column groups in proportion to their tiles, and the slots of one group split the XCD's
// eighth of the row panels, so the re-reads of A by the other column groups hit that XCD's L2.
#pragma once
// (included by gemm_bx.hpp after k_gemm_bxp: uses its split helpers, BxPacked and the PanelBatch / Epi contract)

namespace temp {

#ifdef BXR_PROBE
__device__ unsigned long long g_bxr_stamp[256 * 8 * 4];
__device__ unsigned long long g_bxr_epi[256 * 8];
__device__ unsigned long long g_bxr_rt[256 * 8 * 2];       // s_memrealtime (100 MHz) at the wave's start and end: shader clock = cycles / time
#endif

#define BXR_G 4                                              // column tiles resident per workgroup
#define BXR_WAVES 8
#define BXR_MAX_SLABS 13
#define BXR_BIAS_BYTES (BXR_G * 32 * 4)                      // the group's bias columns behind the planes
#define BXR_LDS_BYTES (BXR_MAX_SLABS * BXR_G * 192 * 16 + BXR_BIAS_BYTES)   // 159 744 + 512 of the CU's 163 840

struct BxrGeom {
  int N, K, lda, n_tiles, n_slabs, n_groups;
  int ldb, trans_b;                                          // B as the caller stores it (blocks split their own slice: no pack launch)
  int per_xcd;                                               // 32-row panels per XCD (of the largest problem)
  unsigned char slot_group[32], slot_rank[32], group_slots[8];
  unsigned char group_t0[8], group_nt[8];                    // first column tile and tile count (<= BXR_G) of every group
};

inline bool bxr_plan(int N, int K, int lda, int max_m, BxrGeom* g) {
  if (K % 8 || lda % 4 || N % 4 || K < 72 || K > BXR_MAX_SLABS * 16) return false;   // >= 5 slabs: the A stream runs four slabs ahead
  g->N = N; g->K = K; g->lda = lda;
  g->n_tiles = ceil_div(N, 32);
  g->n_slabs = ceil_div(K, 16);
  g->n_groups = ceil_div(g->n_tiles, BXR_G);
  if (g->n_groups > 8) return false;
  g->per_xcd = ceil_div(ceil_div(max_m, 32), 8);
  // 32 slots of an XCD -> column groups, in proportion to the groups' tiles (largest remaining load per slot first)
  int tiles[8], slots[8], left = 32;
  for (int j = 0; j < g->n_groups; ++j) {
    tiles[j] = (j + 1 < g->n_groups) ? BXR_G : g->n_tiles - BXR_G * (g->n_groups - 1);
    slots[j] = 1;
    --left;
  }
  while (left > 0) {
    int best = 0;
    for (int j = 1; j < g->n_groups; ++j)
      if ((long long)tiles[j] * slots[best] > (long long)tiles[best] * slots[j]) best = j;
    ++slots[best];
    --left;
  }
  int s = 0;
  for (int j = 0, t = 0; j < g->n_groups; ++j) {
    g->group_t0[j] = (unsigned char)t; g->group_nt[j] = (unsigned char)tiles[j]; t += tiles[j];
    g->group_slots[j] = (unsigned char)slots[j];
    for (int r = 0; r < slots[j]; ++r, ++s) { g->slot_group[s] = (unsigned char)j; g->slot_rank[s] = (unsigned char)r; }
  }
  return true;
}

// An epilogue whose fin4(acc, pre) is  store(f(acc + pre))  with pre4 independent of acc may specialise EpiAccInit to true: its
// pre4 values are then loaded straight INTO the accumulators at the start of a panel (no registers of their own -- 64 at four
// tiles) and fin4 gets a zero `pre`.  (Explicit specialisations only: a derived epilogue does not inherit it.)
template <class E>
struct EpiAccInit { static constexpr bool value = false; };
// An epilogue that declares `k_raw_pre` (has_addend / raw4 / bias1, RowCtx::add, > 0 = keep) has  pre4 = (add ? raw4 : 0) + bias  taken apart:
// the raw loads of a panel are issued back to back, the mask select follows them, the bias comes from LDS (staged once per block).
// pre4 itself branches on kernel-uniform pointers per quad, and the compiler serialises its loads: 32 dependent round trips per
// panel, two thirds of the first version's run time.
template <class E, class = void>
struct EpiRawPre { static constexpr bool value = false; };
template <class E>
struct EpiRawPre<E, decltype((void)E::k_raw_pre)> { static constexpr bool value = true; };

// One wave, GT resident column tiles: the panels [p_lo + first, p_hi) step `stride` of problem `pb`.
//
// The A stream is a ring of FOUR raw register stages with static names (the slab loop is unrolled by four; stage = flat slab
// index mod 4): body f issues the loads of flat slab f + 4 into the stage whose content it no longer needs and splits slab f + 1
// for the next body, so a load has three slab bodies (~2 300 MFMA cycles of this wave alone) before its first use.  The first
// version rotated three stages through register copies (r1 <- r2 <- r3): a copy of an in-flight load makes the wave wait for
// it, so every load was waited for at the end of the body that issued it -- one slab of cover against an HBM latency of 2-3 --
// and the matrix pipe idled two thirds of the time (86-112 us per launch at the S-gdelt shapes).  The operand fragments
// alternate between two named sets for the same reason (no AH <- NH copies).  The flat sequence runs over panel boundaries:
// the next panel's first slabs are in flight during this one's epilogue.
template <int GT, class Epi, int VAR = 0>
__device__ __forceinline__ void bxr_wave(const PanelProblem<Epi>& pb, const BxrGeom& g, const bx_u32x4* __restrict__ Bl, const float* __restrict__ bias_l, int t0, int first, int p_hi, int stride) {
  const int M = pb.M, N = g.N, K = g.K, NS = g.n_slabs;
  const float* __restrict__ A = pb.A;
  const int32_t* __restrict__ a_idx = pb.a_idx;
  const Epi& epi = pb.epi;
  const int lane = threadIdx.x & 63;
  const int hh = lane >> 5, li = lane & 31;
  const int n0 = t0 * 32;
  const int kclamp = K - 8;                                   // last octet that may be read
  if (first >= p_hi) return;

  auto row_src = [&](int panel) {                            // source row of this lane in `panel` (-1: past M, or a gathered zero row)
    const int row = panel * 32 + li;
    int a_src = -1;
    if (panel < p_hi && row < M) a_src = a_idx ? a_idx[row] : row;
    return a_src;                                             // (kept as the loaded word: a widening here would wait for the load)
  };
  auto src_ptr = [&](int a_src) {                            // rows past M / gathered zero rows compute on row 0
    return A + (size_t)(a_src >= 0 ? a_src : 0) * g.lda + 8 * hh;
  };
  auto fetch_a = [&](float4 (&a)[2], const float* aptr, int s) {
    const int k = 16 * s + 8 * hh;
    const float* p = aptr + (k <= kclamp ? 16 * s : kclamp - 8 * hh);   // past K: a valid octet again (meets the zero padding of B)
    if constexpr (VAR & 1) { a[0] = make_float4(1.f, 2.f, 3.f, (float)s); a[1] = a[0]; return; }
    if constexpr (VAR & 16) {                                 // (probe) same bytes, row-contiguous: 4 lanes x 16 B per row and instruction
      const float* q = aptr - 8 * hh + ((lane >> 2) - li) * (long)g.lda + 16 * min(s, NS - 2) + 4 * (lane & 3);
      if (q < A) q = A + 4 * (lane & 3);
      a[0] = ld4(q);
      a[1] = ld4(q + 16 * (long)g.lda);
      return;
    }
    a[0] = ld4(p);
    a[1] = ld4(p + 4);
  };

  constexpr bool INIT = EpiAccInit<Epi>::value, RAW = EpiRawPre<Epi>::value;
  static_assert(!INIT || RAW, "an accumulator-start epilogue supplies the raw-load interface");
  f32x16 acc[GT];
  auto acc_start = [&](int panel, const typename Epi::RowCtx& rc) {   // accumulators of a panel: zero, or the epilogue's addend (INIT)
    const int row = panel * 32 + li;
    const bool row_ok = panel < p_hi && row < M;
    bool loaded = false;
    if constexpr (INIT) {
      if (!(VAR & (2 | 128)) && epi.has_addend()) {           // (kernel-uniform; probe bit 128: no addend loads)
        loaded = true;
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = n0 + t * 32 + 8 * q + 4 * hh;
            const bool ok = row_ok && col < N;
            float4 v;
            if constexpr (VAR & 32) {                         // (probe) same bytes, 8 lanes x 16 B per row and instruction
              const int prow = min(panel * 32 + 8 * q + (lane >> 3), M - 1);
              v = epi.raw4(prow, min(n0 + t * 32, N - 32) + 4 * (lane & 7));
            } else
              v = epi.raw4(ok ? row : 0, ok ? col : 0);
            acc[t][4 * q] = v.x; acc[t][4 * q + 1] = v.y; acc[t][4 * q + 2] = v.z; acc[t][4 * q + 3] = v.w;
          }
        // Wait for them HERE (vmcnt(0): one exposed latency per panel, covered by the SIMD's other wave).  Left to the compiler, the
        // waits land in front of the slab body's MFMAs, where they must also hold on the path that comes from the previous body --
        // with the counter in order that makes every body wait for all but its newest A load, i.e. no prefetch at all.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (__builtin_amdgcn_ballot_w64(!(rc.add > 0)) != 0) {      // a masked row in this panel: its addend is zero
#pragma unroll
          for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = rc.add > 0 ? acc[t][r] : 0.f;
        }
      }
    }
    if (!loaded) {
#pragma unroll
      for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
  };
  int panel = first, s = 0;
  bool row_ok = panel * 32 + li < M;
  typename Epi::RowCtx rc = epi.row_ctx(row_ok ? panel * 32 + li : 0);
  acc_start(first, rc);

  // ---- fetch cursor: (panel pointer, slab) of the next flat slab to load; it runs at most one panel ahead, and wraps into the next panel
  // only after the panel advance below has set src_nxt (NS >= 5)
  int src_cur = row_src(first), src_nxt = row_src(first + stride);    // (a gather index is loaded here and first used at the
  const float* fptr = src_ptr(src_cur);                                //  cursor's wrap, NS - 4 bodies later: no wait for it)
  int fs = 0;
  auto fetch_next = [&](float4 (&a)[2]) {
    fetch_a(a, fptr, fs);
    if (++fs == NS) { fs = 0; fptr = src_ptr(src_nxt); }
  };
  float4 R[4][2];                                             // raw A of flat slabs f .. f + 3 (stage = flat index mod 4)
  bx_u32x4 F[2][3];                                           // operand fragments (h, m, l) of the current / the next flat slab
#pragma unroll
  for (int j = 0; j < 4; ++j) fetch_next(R[j]);
  bx_split8(R[0][0], R[0][1], F[0][0], F[0][1], F[0][2]);
  if (src_cur < 0) { F[0][0] = bx_u32x4{0, 0, 0, 0}; F[0][1] = F[0][0]; F[0][2] = F[0][0]; }   // a gathered zero row (or past M)

  float4 pre[INIT ? 1 : GT][4];
  typename Epi::RowCtx rc_next = rc;                          // the next panel's row context (its mask load), fetched two slabs early
  bool row_masked = false;                                    // (kernel-uniform; without a row mask the context is the same for every row)
  if constexpr (RAW) row_masked = epi.has_row_mask();
  for (;;) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {                             // flat slab f with f mod 4 == j
      const int row = panel * 32 + li;
      const bool zr = (s + 1 < NS) ? (src_cur < 0) : (src_nxt < 0);   // the row the NEXT flat slab belongs to: a zero row?
      fetch_next(R[j]);                                       // flat slab f + 4 (stage j held slab f: split by the previous body)
      if constexpr (RAW) {
        if (s == NS - 2 && row_masked) {
          const int nrow = (panel + stride) * 32 + li;
          rc_next = epi.row_ctx(panel + stride < p_hi && nrow < M ? nrow : 0);
        }
      }
      if (!INIT && s == NS - 2) {                             // the epilogue's own loads: behind the last slabs of the panel
        bool raw_loads = false;
        if constexpr (RAW) raw_loads = epi.has_addend();      // (kernel-uniform)
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = n0 + t * 32 + 8 * q + 4 * hh;
            const bool ok = row_ok && col < N;
            if constexpr (RAW) pre[t][q] = raw_loads ? epi.raw4(ok ? row : 0, ok ? col : 0) : zero4();
            else pre[t][q] = epi.pre4(rc, ok ? row : 0, ok ? col : 0);
          }
      }
      bx_u32x4 (&CF)[3] = F[j & 1];
      bx_u32x4 (&NF)[3] = F[(j + 1) & 1];
      const float4 (&rn)[2] = R[(j + 1) & 3];
      auto chunk = [&](int c) {                               // element pair c of the NEXT slab's fragment
        const float4 f = rn[c >> 1];
        unsigned h, m, l;
        if constexpr (VAR & 8) { h = __float_as_uint(f.x); m = __float_as_uint(f.y); l = __float_as_uint(f.z); }
        else bx_split_pair((c & 1) ? f.z : f.x, (c & 1) ? f.w : f.y, h, m, l);
        NF[0][c] = zr ? 0u : h; NF[1][c] = zr ? 0u : m; NF[2][c] = zr ? 0u : l;
      };
      const bx_bf16x8 ah = bx_frag(CF[0]), am = bx_frag(CF[1]), al = bx_frag(CF[2]);
      const bx_u32x4* bs = Bl + (size_t)s * (BXR_G * 192) + lane;
      // Tiles in PAIRS (the products of tile t alternate with those of tile t + 1: an MFMA never waits for the accumulator of the
      // one just issued), plane by plane -- L.ah | M.am, M.ah | H.al, H.am, H.ah (small terms first within a plane; the order of the
      // window-chain kernels) -- so that a plane's registers are free after its last product and are refilled IN PLACE with the
      // next pair's plane (>= 6 MFMAs = 190 cycles before its first use: covers the LDS latency with a single buffer: 24 registers).
      // Every second MFMA is followed by one element pair of the next slab's operand split.
      constexpr int NP = (GT + 1) / 2;
      bx_u32x4 wf[2][3];                                        // [tile of the pair][plane h, m, l]
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (u < GT) wf[u][p] = bs[(u * 3 + p) * 64];
      __builtin_amdgcn_sched_barrier(0);
      int slot = 0;
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        const bool two = 2 * pr + 1 < GT;
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const int t = 2 * pr + u;
            const bx_bf16x8 wh = bx_frag(wf[u][0]), wm = bx_frag(wf[u][1]), wl = bx_frag(wf[u][2]);
            if (jj == 0 && !(VAR & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc[t], 0, 0, 0);
            if (jj == 1 && !(VAR & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, acc[t], 0, 0, 0);
            if (jj == 2 && !(VAR & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, acc[t], 0, 0, 0);
            if (jj == 3 && !(VAR & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc[t], 0, 0, 0);
            if (jj == 4 && !(VAR & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, acc[t], 0, 0, 0);
            if (jj == 5 && !(VAR & 4)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc[t], 0, 0, 0);
            const int tn = 2 * (pr + 1) + u;                  // the same tile slot of the next pair: its planes as they fall free
            if (pr + 1 < NP && tn < GT) {
              if (jj == 0) wf[u][2] = bs[(tn * 3 + 2) * 64];
              if (jj == 2) wf[u][1] = bs[(tn * 3 + 1) * 64];
              if (jj == 5) wf[u][0] = bs[(tn * 3 + 0) * 64];
            }
            if ((slot & 1) && (slot >> 1) < 4) chunk(slot >> 1);
            ++slot;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#pragma unroll
      for (int c = (GT * 6) >> 1; c < 4; ++c) chunk(c);        // narrow groups: what found no slot behind an MFMA
      // the next body is another basic block (the panel-end branch below): without a use here the compiler SINKS the split of the
      // next fragment out of the MFMA shadow into the top of that block
      asm volatile("" : "+v"(NF[0]), "+v"(NF[1]), "+v"(NF[2]));

      if (++s == NS) {
#ifdef BXR_PROBE
        const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- epilogue of the panel (its own loads are in `pre`, or were the accumulators' start values), then the next panel's start
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = n0 + t * 32 + 8 * q + 4 * hh;
            float4 p = INIT ? zero4() : pre[INIT ? 0 : t][q];
            if constexpr (RAW) {
              if constexpr (!INIT) p = rc.add > 0 ? p : zero4();
              p = add4(p, *reinterpret_cast<const float4*>(bias_l + t * 32 + 8 * q + 4 * hh));
            }
            if constexpr (VAR & 32) {
              const int prow = min(panel * 32 + 8 * q + (lane >> 3), M - 1);
              epi.fin4(rc, prow, min(n0 + t * 32, N - 32) + 4 * (lane & 7), make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]), p);
            } else
            if ((VAR & 64) && acc[t][4 * q] != 1.2345e30f) continue;      // (probe) no stores, accumulators stay live
            if (row_ok && col < N && !(VAR & 2)) epi.fin4(rc, row, col, make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]), p);
          }
        panel += stride;
        if (panel >= p_hi) return;
        s = 0;
        row_ok = panel * 32 + li < M;
        rc = rc_next;
        src_cur = src_nxt;                                    // (the fetch cursor crossed into this panel NS - 4 bodies ago and
        src_nxt = row_src(panel + stride);                    //  holds its pointer; it wraps to src_nxt's after this update: NS >= 5)
        acc_start(panel, rc);                                 // (after the index load: the compiler waits for that one at the block's
                                                              //  end, and the wait inside acc_start then covers it)
#ifdef BXR_PROBE
        if (lane == 0) g_bxr_epi[(size_t)blockIdx.x * BXR_WAVES + (threadIdx.x >> 6)] += __builtin_amdgcn_s_memtime() - te0;
#endif
      }
    }
  }
}

// An epilogue may limit the COLUMNS of B its problem really has (`int n_valid`, <= the launch's N): the problems of one launch
// then share N = the widest and a block never reads rows of B (trans_b = 1) past its own problem's (EpiPlainStoreT: the loss's
// score products written transposed, gemm_kernels.hip).
template <class Epi, class = void> struct EpiColLimit { static __device__ __forceinline__ int get(const Epi&, int n) { return n; } };
template <class Epi> struct EpiColLimit<Epi, decltype((void)Epi::has_n_valid)> {
  static __device__ __forceinline__ int get(const Epi& e, int n) { return e.n_valid < n ? e.n_valid : n; }
};

// VAR is 0 in the library; tools/bxr_probe.hip instantiates ablations (bit0: no A loads, bit1: no epilogue traffic, bit2: no MFMAs,
// bit3: no operand split) and, with BXR_PROBE defined, s_memtime stamps per wave
template <class Epi, int VAR = 0>
__global__ void __launch_bounds__(BXR_WAVES * 64, 2) k_gemm_bxr(PanelBatch<Epi> batch, BxrGeom g, BxPacked packed) {
  extern __shared__ __attribute__((aligned(16))) bx_u32x4 bxr_lds[];
  const PanelProblem<Epi>& pb = batch.p[blockIdx.y];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = g.slot_group[slot], rank = g.slot_rank[slot], nslots = g.group_slots[grp];
  const int t0 = g.group_t0[grp];
  const int gt = g.group_nt[grp];
  const int n_panels = (pb.M + 31) >> 5;
  const int p_lo = xcd * g.per_xcd, p_hi = min(n_panels, p_lo + g.per_xcd);
  if (p_lo >= p_hi) return;                                   // (uniform) nothing for this XCD in this problem
  // ---- the group's planes of B for all of K: slab s = gt * 192 consecutive 16-byte pieces of the packed matrix (<= 2 per thread);
  // four slabs = up to eight loads in flight per thread (a load-wait-store loop of 20 dependent L2 round trips cost ~25 us)
  if (packed.b[blockIdx.y] == nullptr) {
    // no packed copy: the block cuts its OWN slice of B (<= 4 tiles x 13 slabs x 64 sixteen-byte items = 6.5 items per thread, 36
    // VALU instructions each) straight from the fp32 matrix -- the pack launch in front of every weights-resident product is gone
    const float* __restrict__ B = pb.B;
    const int items = g.n_slabs * gt * 64, ldb = g.ldb;
    const int n_lim = EpiColLimit<Epi>::get(pb.epi, g.N);
    constexpr int UN = 4;
    for (int i0 = threadIdx.x; i0 < items; i0 += UN * BXR_WAVES * 64) {
      float4 v0[UN], v1[UN];
      int dst[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int it = i0 + u * BXR_WAVES * 64;
        const int itc = min(it, items - 1);
        const int ln = itc & 63, sj = itc >> 6, sl = sj / gt, j = sj - sl * gt;
        const int k = 16 * sl + 8 * (ln >> 5), n = (t0 + j) * 32 + (ln & 31);
        dst[u] = it < items ? sl * (BXR_G * 192) + j * 192 + ln : -1;
        v0[u] = zero4(); v1[u] = zero4();
        if (n < n_lim && k < g.K) {                             // K % 8 == 0: the octet is entirely in or out
          if (g.trans_b) {
            const float* q = B + (size_t)n * ldb + k;
            v0[u] = ld4(q); v1[u] = ld4(q + 4);
          } else {
            const float* q = B + (size_t)k * ldb + n;
            const size_t l = (size_t)ldb;
            v0[u] = make_float4(q[0], q[l], q[2 * l], q[3 * l]);
            v1[u] = make_float4(q[4 * l], q[5 * l], q[6 * l], q[7 * l]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (dst[u] < 0) continue;
        bx_u32x4 H, Mi, L;
        bx_split8(v0[u], v1[u], H, Mi, L);
        bx_u32x4* d = bxr_lds + dst[u];
        d[0] = H; d[64] = Mi; d[128] = L;
      }
    }
  } else {
    const bx_u32x4* __restrict__ src = packed.b[blockIdx.y] + (size_t)t0 * 192;
    const int per = gt * 192, NS = g.n_slabs;
    const size_t sstride = (size_t)g.n_tiles * 192;
    const int r0 = min((int)threadIdx.x, per - 1), r1 = min((int)threadIdx.x + BXR_WAVES * 64, per - 1);
    const bool ok0 = (int)threadIdx.x < per, ok1 = (int)threadIdx.x + BXR_WAVES * 64 < per;
    for (int s0 = 0; s0 < NS; s0 += 4) {
      bx_u32x4 v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bx_u32x4* p = src + (size_t)min(s0 + u, NS - 1) * sstride;
        v[u][0] = p[r0];
        v[u][1] = p[r1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u < NS) {
          bx_u32x4* d = bxr_lds + (s0 + u) * (BXR_G * 192);
          if (ok0) d[r0] = v[u][0];
          if (ok1) d[r1] = v[u][1];
        }
    }
  }
#ifdef BXR_PROBE
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
  const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif
  float* bias_l = reinterpret_cast<float*>(bxr_lds + g.n_slabs * (BXR_G * 192));
  if constexpr (EpiRawPre<Epi>::value) {
    if (threadIdx.x < BXR_G * 32) {
      const int col = t0 * 32 + (int)threadIdx.x;
      bias_l[threadIdx.x] = col < g.N ? pb.epi.bias1(col) : 0.f;
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int first = p_lo + rank * BXR_WAVES + wave, stride = nslots * BXR_WAVES;
#ifdef BXR_PROBE
  const unsigned long long t_staged = __builtin_amdgcn_s_memtime();
#endif
  // Whole rounds of panels go to the waves with all gt tiles each; the LAST, partial round (r < stride panels) is cut into r x gt
  // single-tile units dealt round the waves: with 2.2 panels per wave (the self-loop products of the step) every wave used to
  // wait for the few that ran a third panel -- 12 tile-panels on the critical path against an average of 8.8; now 8 + 1.
  // (only where a wave gets at most ONE unit: a unit pays a pipeline fill, an epilogue and a full operand split for six MFMAs per
  //  slab, three of them cost more than the panel they replace -- measured on the input-gate shape, +5 %)
  const int n_x = p_hi - p_lo;
  int q_full = n_x / stride, n_units = (n_x - q_full * stride) * gt;
  if (n_units > stride) { q_full = (n_x + stride - 1) / stride; n_units = 0; }
  const int p_full = min(p_hi, p_lo + q_full * stride);
  if (q_full > 0) {
    if (gt == 4) bxr_wave<4, Epi, VAR>(pb, g, bxr_lds, bias_l, t0, first, p_full, stride);
    else if (gt == 3) bxr_wave<3, Epi, VAR>(pb, g, bxr_lds, bias_l, t0, first, p_full, stride);
    else if (gt == 2) bxr_wave<2, Epi, VAR>(pb, g, bxr_lds, bias_l, t0, first, p_full, stride);
    else bxr_wave<1, Epi, VAR>(pb, g, bxr_lds, bias_l, t0, first, p_full, stride);
  }
  for (int u = rank * BXR_WAVES + wave; u < n_units; u += stride) {
    const int panel = p_full + u / gt, ti = u - (u / gt) * gt;
    bxr_wave<1, Epi, VAR>(pb, g, bxr_lds + ti * 192, bias_l + ti * 32, t0 + ti, panel, panel + 1, 1);
  }
#ifdef BXR_PROBE
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* o = g_bxr_stamp + ((size_t)blockIdx.x * BXR_WAVES + wave) * 4;
    o[0] = t_start; o[1] = t_staged; o[2] = __builtin_amdgcn_s_memtime();
    g_bxr_rt[((size_t)blockIdx.x * BXR_WAVES + wave) * 2] = rt_start;
    g_bxr_rt[((size_t)blockIdx.x * BXR_WAVES + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    o[3] = ((unsigned long long)gt << 32) | (unsigned)(first < p_hi ? (p_hi - first + stride - 1) / stride : 0);
  }
#endif
}

}  // namespace temp
