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

#define BXR_G 4                                              // column tiles resident per workgroup
#define BXR_WAVES 8
#define BXR_MAX_SLABS 13
#define BXR_LDS_BYTES (BXR_MAX_SLABS * BXR_G * 192 * 16)     // 159 744

struct BxrGeom {
  int N, K, lda, n_tiles, n_slabs, n_groups;
  int per_xcd;                                               // 32-row panels per XCD (of the largest problem)
  unsigned char slot_group[32], slot_rank[32], group_slots[8];
};

inline bool bxr_plan(int N, int K, int lda, int max_m, BxrGeom* g) {
  if (K % 8 || lda % 4 || N % 4 || K < 48 || K > BXR_MAX_SLABS * 16) return false;   // >= 3 slabs: the A stream runs three slabs ahead
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
  for (int j = 0; j < g->n_groups; ++j) {
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

// One wave, GT resident column tiles: the panels [p_lo + first, p_hi) step `stride` of problem `pb`.
template <int GT, class Epi>
__device__ __forceinline__ void bxr_wave(const PanelProblem<Epi>& pb, const BxrGeom& g, const bx_u32x4* __restrict__ Bl, int t0, int first, int p_hi, int stride) {
  const int M = pb.M, N = g.N, K = g.K, NS = g.n_slabs;
  const float* __restrict__ A = pb.A;
  const int32_t* __restrict__ a_idx = pb.a_idx;
  const Epi& epi = pb.epi;
  const int lane = threadIdx.x & 63;
  const int hh = lane >> 5, li = lane & 31;
  const int n0 = t0 * 32;
  const int kclamp = K - 8;                                   // last octet that may be read
  if (first >= p_hi) return;

  auto row_ptr = [&](int panel, long& a_src) {
    const int row = panel * 32 + li;
    a_src = -1;
    if (panel < p_hi && row < M) a_src = a_idx ? (long)a_idx[row] : (long)row;
    return A + (size_t)(a_src >= 0 ? a_src : 0) * g.lda + 8 * hh;   // rows past M / gathered zero rows compute on row 0
  };
  auto fetch_a = [&](float4 (&a)[2], const float* aptr, int s) {
    const int k = 16 * s + 8 * hh;
    const float* p = aptr + (k <= kclamp ? 16 * s : kclamp - 8 * hh);   // past K: a valid octet again (meets the zero padding of B)
    a[0] = ld4(p);
    a[1] = ld4(p + 4);
  };

  constexpr bool INIT = EpiAccInit<Epi>::value;
  f32x16 acc[GT];
  auto acc_start = [&](int panel) {                           // accumulators of a panel: zero, or the epilogue's addend (INIT)
    const int row = panel * 32 + li;
    const bool row_ok = panel < p_hi && row < M;
    typename Epi::RowCtx rc;
    if constexpr (INIT) rc = epi.row_ctx(row_ok ? row : 0);
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = zero4();
        if constexpr (INIT) {
          const int col = n0 + t * 32 + 8 * q + 4 * hh;
          const bool ok = row_ok && col < N;
          v = epi.pre4(rc, ok ? row : 0, ok ? col : 0);
        }
        acc[t][4 * q] = v.x; acc[t][4 * q + 1] = v.y; acc[t][4 * q + 2] = v.z; acc[t][4 * q + 3] = v.w;
      }
  };
  acc_start(first);

  // ---- A stream: slab s of the current panel is split one slab ahead (NH/NM/NL), its raw octets arrive three slabs ahead.  The
  // flat sequence (panel, slab) runs over panel boundaries: the next panel's first slabs are in flight during this one's epilogue.
  long src_cur, src_nxt;
  const float* ptr_cur = row_ptr(first, src_cur);
  const float* ptr_nxt = row_ptr(first + stride, src_nxt);
  float4 r1[2], r2[2], r3[2];                                 // raw A of flat slabs +1, +2, +3
  bx_u32x4 AH, AM, AL, NH, NM, NL;
  {
    float4 r0[2];
    fetch_a(r0, ptr_cur, 0);
    fetch_a(r1, NS > 1 ? ptr_cur : ptr_nxt, NS > 1 ? 1 : 0);
    fetch_a(r2, NS > 2 ? ptr_cur : ptr_nxt, NS > 2 ? 2 : (NS > 1 ? 0 : 1));
    bx_split8(r0[0], r0[1], AH, AM, AL);
  }
  if (src_cur < 0) { AH = bx_u32x4{0, 0, 0, 0}; AM = AH; AL = AH; }     // a gathered zero row (or a row past M): zero operand
  auto chunk = [&](int c, bool zero_row) {                    // element pair c of the NEXT slab's fragment
    const float4 f = r1[c >> 1];
    unsigned h, m, l;
    bx_split_pair((c & 1) ? f.z : f.x, (c & 1) ? f.w : f.y, h, m, l);
    NH[c] = zero_row ? 0u : h; NM[c] = zero_row ? 0u : m; NL[c] = zero_row ? 0u : l;
  };

  for (int panel = first; panel < p_hi; panel += stride) {
    const int row = panel * 32 + li;
    const bool row_ok = row < M;
    const typename Epi::RowCtx rc = epi.row_ctx(row_ok ? row : 0);
    float4 pre[INIT ? 1 : GT][4];
    for (int s = 0; s < NS; ++s) {
      const bool zr = (s + 1 < NS) ? (src_cur < 0) : (src_nxt < 0);   // the row the NEXT flat slab belongs to: a zero row?
      // flat slab s + 3: this panel's or the next one's
      {
        const int f = s + 3;
        const bool here = f < NS;
        fetch_a(r3, here ? ptr_cur : ptr_nxt, here ? f : f - NS);
      }
      if (!INIT && s == NS - 2) {                             // the epilogue's own loads: behind the last slabs of the panel
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = n0 + t * 32 + 8 * q + 4 * hh;
            const bool ok = row_ok && col < N;
            pre[t][q] = epi.pre4(rc, ok ? row : 0, ok ? col : 0);
          }
      }
      const bx_bf16x8 ah = bx_frag(AH), am = bx_frag(AM), al = bx_frag(AL);
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
        for (int j = 0; j < 6; ++j) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const int t = 2 * pr + u;
            const bx_bf16x8 wh = bx_frag(wf[u][0]), wm = bx_frag(wf[u][1]), wl = bx_frag(wf[u][2]);
            if (j == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc[t], 0, 0, 0);
            if (j == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, acc[t], 0, 0, 0);
            if (j == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, acc[t], 0, 0, 0);
            if (j == 3) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc[t], 0, 0, 0);
            if (j == 4) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, acc[t], 0, 0, 0);
            if (j == 5) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc[t], 0, 0, 0);
            const int tn = 2 * (pr + 1) + u;                  // the same tile slot of the next pair: its planes as they fall free
            if (pr + 1 < NP && tn < GT) {
              if (j == 0) wf[u][2] = bs[(tn * 3 + 2) * 64];
              if (j == 2) wf[u][1] = bs[(tn * 3 + 1) * 64];
              if (j == 5) wf[u][0] = bs[(tn * 3 + 0) * 64];
            }
            if ((slot & 1) && (slot >> 1) < 4) chunk(slot >> 1, zr);
            ++slot;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#pragma unroll
      for (int c = (GT * 6) >> 1; c < 4; ++c) chunk(c, zr);     // narrow groups: what found no slot behind an MFMA
      AH = NH; AM = NM; AL = NL;
      r1[0] = r2[0]; r1[1] = r2[1];
      r2[0] = r3[0]; r2[1] = r3[1];
    }
    // ---- epilogue of the panel (its own loads are in `pre`, or were the accumulators' start values), then the next panel's start
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + t * 32 + 8 * q + 4 * hh;
        if (row_ok && col < N) epi.fin4(rc, row, col, make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]),
                                        INIT ? zero4() : pre[INIT ? 0 : t][q]);
      }
    acc_start(panel + stride);
    ptr_cur = ptr_nxt;
    src_cur = src_nxt;
    ptr_nxt = row_ptr(panel + 2 * stride, src_nxt);
  }
}

template <class Epi>
__global__ void __launch_bounds__(BXR_WAVES * 64, 2) k_gemm_bxr(PanelBatch<Epi> batch, BxrGeom g, BxPacked packed) {
  extern __shared__ __attribute__((aligned(16))) bx_u32x4 bxr_lds[];
  const PanelProblem<Epi>& pb = batch.p[blockIdx.y];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = g.slot_group[slot], rank = g.slot_rank[slot], nslots = g.group_slots[grp];
  const int t0 = grp * BXR_G;
  const int gt = min(BXR_G, g.n_tiles - t0);
  const int n_panels = (pb.M + 31) >> 5;
  const int p_lo = xcd * g.per_xcd, p_hi = min(n_panels, p_lo + g.per_xcd);
  if (p_lo >= p_hi) return;                                   // (uniform) nothing for this XCD in this problem
  // ---- the group's planes of B for all of K: slab s = gt * 192 consecutive 16-byte pieces of the packed matrix
  {
    const bx_u32x4* __restrict__ src = packed.b[blockIdx.y] + (size_t)t0 * 192;
    const int per = gt * 192, total = g.n_slabs * per;
    for (int i = threadIdx.x; i < total; i += BXR_WAVES * 64) {
      const int s = i / per, r = i - s * per;
      bxr_lds[s * (BXR_G * 192) + r] = src[(size_t)s * g.n_tiles * 192 + r];
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int first = p_lo + rank * BXR_WAVES + wave, stride = nslots * BXR_WAVES;
  if (gt == 4) bxr_wave<4, Epi>(pb, g, bxr_lds, t0, first, p_hi, stride);
  else if (gt == 3) bxr_wave<3, Epi>(pb, g, bxr_lds, t0, first, p_hi, stride);
  else if (gt == 2) bxr_wave<2, Epi>(pb, g, bxr_lds, t0, first, p_hi, stride);
  else bxr_wave<1, Epi>(pb, g, bxr_lds, t0, first, p_hi, stride);
}

}  // namespace temp
