// gemm_bxr.hpp on the f16 matrix pipe: weights RESIDENT in LDS as two f16 planes (scaled two-way split, split_f16.hpp), every wave
// streams 32-row panels of A through them with three MFMA products per tile and slab where the bf16 kernel issues six.
//
//   C[M, N] = epi( A[M, K] . B ),   K <= 208 (13 slabs of 16), M in the tens of thousands.
//
// What changes against k_gemm_bxr:
//  * The workgroup cuts its OWN slice of B (<= 4 column tiles x all of K) into the planes in two steps: the slice is loaded once into
//    registers, the column maxima are taken through LDS (integer maxima), then every element is scaled by its column's power of two,
//    split and stored -- 104 KB of planes instead of 156 KB.  The inverse column scales sit in LDS beside the bias.
//  * A row's scale comes from its key (PanelProblem::a_keys; by output row or by source row, BxrGeom-independent flag): one dword load
//    per panel and lane, issued with the row's gather index a panel ahead.
//  * The epilogue unscales: C = acc / (row scale . column scale).  The addend of the self-loop layer can no longer START the
//    accumulators (they hold scaled sums): it is loaded behind the last slabs of a panel like any other epilogue operand.
// Panel dealing, the four-stage A ring, the pair-wise MFMA order, the single-tile units of the last round: as in gemm_bxr.hpp.
#pragma once
// (included by gemm_bx.hpp after gemm_hx.hpp: uses BxrGeom / bxr_plan, the PanelBatch / Epi contract, EpiRawPre)

namespace temp {

#define HXR_LDS_BYTES (BXR_MAX_SLABS * BXR_G * 128 * 16 + 3 * BXR_BIAS_BYTES)     // planes + bias + inverse column scales + column keys

template <int GT, class Epi>
__device__ __forceinline__ void hxr_wave(const PanelProblem<Epi>& pb, const BxrGeom& g, const hx_u32x4* __restrict__ Bl, const float* __restrict__ bias_l,
                                         const float* __restrict__ inv_l, int t0, int first, int p_hi, int stride, int keys_by_out) {
  const int M = pb.M, N = g.N, K = g.K, NS = g.n_slabs;
  const float* __restrict__ A = pb.A;
  const int32_t* __restrict__ a_idx = pb.a_idx;
  const unsigned* __restrict__ a_keys = pb.a_keys;
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
    return a_src;
  };
  auto src_ptr = [&](int a_src) { return A + (size_t)(a_src >= 0 ? a_src : 0) * g.lda + 8 * hh; };
  auto row_key = [&](int panel, int a_src) {                 // (a zero row: any key -- its fragments are zeroed)
    const int row = panel * 32 + li;
    return a_keys[a_src >= 0 ? (keys_by_out ? row : a_src) : 0];
  };
  auto fetch_a = [&](float4 (&a)[2], const float* aptr, int s) {
    const int k = 16 * s + 8 * hh;
    const float* p = aptr + (k <= kclamp ? 16 * s : kclamp - 8 * hh);   // past K: a valid octet again (meets the zero padding of B)
    a[0] = ld4(p);
    a[1] = ld4(p + 4);
  };

  constexpr bool RAW = EpiRawPre<Epi>::value;
  f32x16 acc[GT];
#pragma unroll
  for (int t = 0; t < GT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  int panel = first, s = 0;
  bool row_ok = panel * 32 + li < M;
  typename Epi::RowCtx rc = epi.row_ctx(row_ok ? panel * 32 + li : 0);

  // ---- fetch cursor: (panel pointer, slab) of the next flat slab to load; it runs at most one panel ahead (NS >= 5)
  int src_cur = row_src(first), src_nxt = row_src(first + stride);
  unsigned key_cur = row_key(first, src_cur), key_nxt = 0;
  float sc_cur = hx_scale(key_cur), sc_nxt = 1.f;
  const float* fptr = src_ptr(src_cur);
  int fs = 0;
  auto fetch_next = [&](float4 (&a)[2]) {
    fetch_a(a, fptr, fs);
    if (++fs == NS) { fs = 0; fptr = src_ptr(src_nxt); }
  };
  float4 R[4][2];                                             // raw A of flat slabs f .. f + 3 (stage = flat index mod 4)
  hx_u32x4 F[2][2];                                           // operand fragments (h, l) of the current / the next flat slab
#pragma unroll
  for (int j = 0; j < 4; ++j) fetch_next(R[j]);
  hx_split8(R[0][0], R[0][1], sc_cur, F[0][0], F[0][1]);
  if (src_cur < 0) { F[0][0] = hx_u32x4{0, 0, 0, 0}; F[0][1] = F[0][0]; }   // a gathered zero row (or past M)

  float4 pre[GT][4];
  typename Epi::RowCtx rc_next = rc;                          // the next panel's row context (its mask load), fetched two slabs early
  bool row_masked = false;
  if constexpr (RAW) row_masked = epi.has_row_mask();
  for (;;) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {                             // flat slab f with f mod 4 == j
      const int row = panel * 32 + li;
      const bool nxt_panel = !(s + 1 < NS);                   // the NEXT flat slab belongs to the next panel
      const bool zr = nxt_panel ? (src_nxt < 0) : (src_cur < 0);
      const float scn = nxt_panel ? sc_nxt : sc_cur;
      fetch_next(R[j]);                                       // flat slab f + 4 (stage j held slab f: split by the previous body)
      if (s == 1) key_nxt = row_key(panel + stride, src_nxt); // (src_nxt was loaded a panel ago: no wait; first use at body NS - 1)
      if (s == NS - 2) sc_nxt = hx_scale(key_nxt);
      if constexpr (RAW) {
        if (s == NS - 2 && row_masked) {
          const int nrow = (panel + stride) * 32 + li;
          rc_next = epi.row_ctx(panel + stride < p_hi && nrow < M ? nrow : 0);
        }
      }
      if (s == NS - 2) {                                      // the epilogue's own loads: behind the last slabs of the panel
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
      hx_u32x4 (&CF)[2] = F[j & 1];
      hx_u32x4 (&NF)[2] = F[(j + 1) & 1];
      const float4 (&rn)[2] = R[(j + 1) & 3];
      auto chunk = [&](int c) {                               // element pair c of the NEXT slab's fragment
        const float4 f = rn[c >> 1];
        unsigned h, l;
        hx_split_pair((c & 1) ? f.z : f.x, (c & 1) ? f.w : f.y, scn, h, l);
        NF[0][c] = zr ? 0u : h; NF[1][c] = zr ? 0u : l;
      };
      const hx_f16x8 ah = hx_frag(CF[0]), al = hx_frag(CF[1]);
      const hx_u32x4* bs = Bl + (size_t)s * (BXR_G * 128) + lane;
      // Tiles in PAIRS, plane by plane -- L.ah | H.al, H.ah -- so that a plane's registers are free after its last product and are
      // refilled IN PLACE with the next pair's plane; every MFMA pair is followed by one element pair of the next slab's split.
      constexpr int NP = (GT + 1) / 2;
      hx_u32x4 wf[2][2];                                        // [tile of the pair][plane h, l]
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          if (u < GT) wf[u][p] = bs[(u * 2 + p) * 64];
      __builtin_amdgcn_sched_barrier(0);
      int slot = 0;
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        const bool two = 2 * pr + 1 < GT;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) continue;
            const int t = 2 * pr + u;
            const hx_f16x8 wh = hx_frag(wf[u][0]), wl = hx_frag(wf[u][1]);
            if (jj == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah, acc[t], 0, 0, 0);
            if (jj == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, acc[t], 0, 0, 0);
            if (jj == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, acc[t], 0, 0, 0);
            const int tn = 2 * (pr + 1) + u;                  // the same tile slot of the next pair: its planes as they fall free
            if (pr + 1 < NP && tn < GT) {
              if (jj == 0) wf[u][1] = bs[(tn * 2 + 1) * 64];
              if (jj == 2) wf[u][0] = bs[(tn * 2 + 0) * 64];
            }
            if ((slot & 1) && (slot >> 1) < 4) chunk(slot >> 1);
            ++slot;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#pragma unroll
      for (int c = (GT * 3) >> 1; c < 4; ++c) chunk(c);        // narrow groups: what found no slot behind an MFMA
      // (without a use here the compiler sinks the split of the next fragment out of the MFMA shadow into the next basic block)
      asm volatile("" : "+v"(NF[0]), "+v"(NF[1]));

      if (++s == NS) {
        // ---- epilogue of the panel, then the next panel's start
        const float ia = hx_inv_scale(key_cur);
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cl = t * 32 + 8 * q + 4 * hh, col = n0 + cl;
            float4 p = pre[t][q];
            if constexpr (RAW) {
              p = rc.add > 0 ? p : zero4();
              p = add4(p, *reinterpret_cast<const float4*>(bias_l + cl));
            }
            const float4 iv = *reinterpret_cast<const float4*>(inv_l + cl);
            if (row_ok && col < N)
              epi.fin4(rc, row, col, make_float4(acc[t][4 * q] * (ia * iv.x), acc[t][4 * q + 1] * (ia * iv.y), acc[t][4 * q + 2] * (ia * iv.z), acc[t][4 * q + 3] * (ia * iv.w)), p);
            acc[t][4 * q] = 0.f; acc[t][4 * q + 1] = 0.f; acc[t][4 * q + 2] = 0.f; acc[t][4 * q + 3] = 0.f;
          }
        panel += stride;
        if (panel >= p_hi) return;
        s = 0;
        row_ok = panel * 32 + li < M;
        rc = rc_next;
        src_cur = src_nxt;                                    // (the fetch cursor crossed into this panel NS - 4 bodies ago and
        src_nxt = row_src(panel + stride);                    //  holds its pointer; it wraps to src_nxt's after this update: NS >= 5)
        key_cur = key_nxt; sc_cur = sc_nxt;
      }
    }
  }
}

template <class Epi>
__global__ void __launch_bounds__(BXR_WAVES * 64, 2) k_gemm_hxr(PanelBatch<Epi> batch, BxrGeom g, int keys_by_out) {
  extern __shared__ __attribute__((aligned(16))) hx_u32x4 hxr_lds[];
  const PanelProblem<Epi>& pb = batch.p[blockIdx.y];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = g.slot_group[slot], rank = g.slot_rank[slot], nslots = g.group_slots[grp];
  const int t0 = g.group_t0[grp];
  const int gt = g.group_nt[grp];
  const int n_panels = (pb.M + 31) >> 5;
  const int p_lo = xcd * g.per_xcd, p_hi = min(n_panels, p_lo + g.per_xcd);
  if (p_lo >= p_hi) return;                                   // (uniform) nothing for this XCD in this problem
  float* bias_l = reinterpret_cast<float*>(hxr_lds + g.n_slabs * (BXR_G * 128));
  float* inv_l = bias_l + BXR_G * 32;
  unsigned* ckey_l = reinterpret_cast<unsigned*>(inv_l + BXR_G * 32);
  if (threadIdx.x < BXR_G * 32) ckey_l[threadIdx.x] = 0u;
  __syncthreads();
  {
    // ---- the group's slice of B: <= 4 tiles x 13 slabs x 64 sixteen-byte items = up to 7 items per thread, held in registers
    // between the two steps (column maxima, then scale + split)
    const float* __restrict__ B = pb.B;
    const int items = g.n_slabs * gt * 64, ldb = g.ldb;
    constexpr int UN = (BXR_MAX_SLABS * BXR_G * 64 + BXR_WAVES * 64 - 1) / (BXR_WAVES * 64);
    float4 v0[UN], v1[UN];
    int dst[UN], ncol[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int it = (int)threadIdx.x + u * BXR_WAVES * 64;
      const int itc = min(it, items - 1);
      const int ln = itc & 63, sj = itc >> 6, sl = sj / gt, j = sj - sl * gt;
      const int k = 16 * sl + 8 * (ln >> 5), n = (t0 + j) * 32 + (ln & 31);
      dst[u] = it < items ? sl * (BXR_G * 128) + j * 128 + ln : -1;
      ncol[u] = j * 32 + (ln & 31);
      v0[u] = zero4(); v1[u] = zero4();
      if (n < g.N && k < g.K) {                               // K % 8 == 0: the octet is entirely in or out
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
    for (int u = 0; u < UN; ++u)
      if (dst[u] >= 0) atomicMax(&ckey_l[ncol[u]], max(hx_abs_bits4(v0[u]), hx_abs_bits4(v1[u])));
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (dst[u] < 0) continue;
      hx_u32x4 H, L;
      hx_split8(v0[u], v1[u], hx_scale(ckey_l[ncol[u]]), H, L);
      hx_u32x4* d = hxr_lds + dst[u];
      d[0] = H; d[64] = L;
    }
  }
  if (threadIdx.x < BXR_G * 32) {
    const int col = t0 * 32 + (int)threadIdx.x;
    float b = 0.f;
    if constexpr (EpiRawPre<Epi>::value) b = col < g.N ? pb.epi.bias1(col) : 0.f;
    bias_l[threadIdx.x] = b;
    inv_l[threadIdx.x] = hx_inv_scale(ckey_l[threadIdx.x]);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int first = p_lo + rank * BXR_WAVES + wave, stride = nslots * BXR_WAVES;
  // whole rounds of panels with all gt tiles; the last, partial round as single-tile units (gemm_bxr.hpp)
  const int n_x = p_hi - p_lo;
  int q_full = n_x / stride, n_units = (n_x - q_full * stride) * gt;
  if (n_units > stride) { q_full = (n_x + stride - 1) / stride; n_units = 0; }
  const int p_full = min(p_hi, p_lo + q_full * stride);
  if (q_full > 0) {
    if (gt == 4) hxr_wave<4, Epi>(pb, g, hxr_lds, bias_l, inv_l, t0, first, p_full, stride, keys_by_out);
    else if (gt == 3) hxr_wave<3, Epi>(pb, g, hxr_lds, bias_l, inv_l, t0, first, p_full, stride, keys_by_out);
    else if (gt == 2) hxr_wave<2, Epi>(pb, g, hxr_lds, bias_l, inv_l, t0, first, p_full, stride, keys_by_out);
    else hxr_wave<1, Epi>(pb, g, hxr_lds, bias_l, inv_l, t0, first, p_full, stride, keys_by_out);
  }
  for (int u = rank * BXR_WAVES + wave; u < n_units; u += stride) {
    const int panel = p_full + u / gt, ti = u - (u / gt) * gt;
    hxr_wave<1, Epi>(pb, g, hxr_lds + ti * 128, bias_l + ti * 32, inv_l + ti * 32, t0 + ti, panel, panel + 1, 1, keys_by_out);
  }
}

}  // namespace temp
