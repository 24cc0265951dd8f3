// Persistent window-chain kernels of the TeMP snapshot encoder on gfx950 (include/temp_amd.h: TempGruChain).
//
// The GRU recurrence over the train-seq-len window (models/DynamicRGCN.py:156-174, models/BiDynamicRGCN.py:51-100,
// GRRGCNLayer.forward models/RRGCN.py:77-89) is independent per (window, direction, entity): position p of an entity
// reads only that entity's state at position p-1.  A workgroup therefore owns a PANEL of 32 entity tracks, keeps the 32
// current states in LDS and walks every position of the chain inside one launch -- no grid barrier, no relaunch, the
// state never leaves the CU.  512 (forward) / 768 (backward) threads per workgroup, one workgroup per CU:
//   waves 0-3  "matrix" role, one per SIMD:  acc[32 x tile] += W_hh-fragment x state-fragment on v_mfma_f32_32x32x2_f32.
//              W_hh arrives straight from L2 in MFMA fragment order (packed once per step by k_gru_chain_pack: one
//              coalesced 1-KB load per wave per 4 MFMAs per tile), the state fragment is one ds_read_b128 per 4 MFMAs of
//              all tiles; the raw 32 x 3d products are handed over through LDS.
//   waves 4+   "memory" role: everything that touches HBM.  The hoisted input gates (forward) / saved gate planes
//              (backward) of the NEXT position are loaded into registers while the matrix waves work on the current one;
//              after the hand-over barrier these waves apply the gates, write the saved planes / gate gradients with
//              row-contiguous float4 stores and put the new state (resp. d_prev) back into LDS.
// The matrix waves never wait on HBM: their only global loads are L2 hits on the packed weights.
#include "common.hpp"
#include "gru_math.hpp"
#include "gemm_bx.hpp"

namespace temp {

#define CH_SLOTS TEMP_CHAIN_TRACKS
#define CH_HAS_PREV TEMP_CHAIN_HAS_PREV
#define CH_ROW_MASK (TEMP_CHAIN_HAS_PREV - 1)
#define CH_LDS_LIMIT (160 * 1024)

struct ChainRnn { const float4* wf; const float4* wb; const float* b_hh; const unsigned* kf; const unsigned* kb; };   // kf / kb: column keys of the f16 planes (gru_chain_hx.hpp)
struct ChainArgs {
  int D, n_panels, max_steps, dbg;
  int n_rnn_keys;                        // GRUs of the chain (rows of the column-key result in front of the per-panel partials)
  const int32_t* panel; const int32_t* rows; const int32_t* sinfo;
  const float* dt;
  const int32_t* gi_index;
  float lambda;
  size_t plane;
  ChainRnn rnn[TEMP_CHAIN_MAX_RNN];
};
struct ChainUps { const float* p[TEMP_CHAIN_MAX_UP]; };

struct ChainGeom {
  int NT, NQ;        // forward: tiles of 32 gate columns (3d), k-steps of 8 (d), even
  int NTb, NQb;      // backward: tiles of 32 state columns (d), k-steps of 8 (3d), multiple of 4
  int lda, ldh;      // forward LDS strides (floats): product rows, state rows
  int ldA, ldz;      // backward LDS strides: dgh rows, gz / d_prev rows
};
__host__ __device__ inline ChainGeom chain_geom(int D) {
  ChainGeom g;
  g.NT = (3 * D + 31) >> 5; g.NQ = ((D + 15) >> 4) << 1;          // k-steps come in groups (the loops are unrolled by two,
  g.NTb = (D + 31) >> 5; g.NQb = ((3 * D + 31) >> 5) << 2;        // by four in the backward, branch-free): padded with zero stages
  g.lda = g.NT * 32 + 4; g.ldh = g.NQ * 8 + 4;          // (stride / 4) odd: conflict-free ds_read_b128 / ds_write_b128 across rows
  g.ldA = g.NQb * 8 + 4; g.ldz = g.NTb * 32 + 4;
  return g;
}
#define CH_MAX_STEPS 64      // steps of one panel: its row table, decay factors and step flags are staged in LDS
// `ms` = the longest panel of the launch (<= CH_MAX_STEPS)
inline size_t chain_lds_fwd(int D, int ms) { ChainGeom g = chain_geom(D); return ((size_t)CH_SLOTS * g.lda + 2 * CH_SLOTS * g.ldh + (2 * CH_SLOTS + 1) * (size_t)ms) * 4; }
inline size_t chain_lds_bwd(int D, int ms) { ChainGeom g = chain_geom(D); return ((size_t)CH_SLOTS * g.ldA + 2 * CH_SLOTS * g.ldz + (2 * CH_SLOTS + 3) * (size_t)ms) * 4; }

}  // namespace temp
#include "gru_chain_hx.hpp"
namespace temp {

// ---- W_hh -> fragment order ---------------------------------------------------------------------------------------
// forward  piece (tile, q, lane): float4 e -> W_hh[tile*32 + li][8q + 4hh + e]      (gate column x k)
// backward piece (tile, q, lane): float4 e -> W_hh[8q + 4hh + e][tile*32 + li]      (k = gate column, state column)
// zero outside the matrix.  li = lane & 31, hh = lane >> 5: the A-operand layout of v_mfma_f32_32x32x2_f32 (lane supplies
// A[i = li][k = hh]); one float4 feeds 4 consecutive MFMAs, the state fragment uses the same k order.
__global__ void __launch_bounds__(256) k_gru_chain_pack(int D, const float* __restrict__ W, float4* __restrict__ out) {
  const ChainGeom g = chain_geom(D);
  const int nf = g.NT * g.NQ * 64, nb = g.NTb * g.NQb * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += gridDim.x * blockDim.x) {
    const bool fwd = i < nf;
    const int j = fwd ? i : i - nf;
    const int lane = j & 63, rest = j >> 6;
    const int nq = fwd ? g.NQ : g.NQb;
    const int tile = rest / nq, q = rest - tile * nq;
    const int n = tile * 32 + (lane & 31), k = 8 * q + 4 * (lane >> 5);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (fwd) v[e] = (n < 3 * D && k + e < D) ? W[(size_t)n * D + k + e] : 0.f;
      else v[e] = (n < D && k + e < 3 * D) ? W[(size_t)(k + e) * D + n] : 0.f;
    }
    out[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---- forward --------------------------------------------------------------------------------------------------------
// TPW = tiles per matrix wave (ceil(NT / 4)), MW = memory waves (4 or 8).
// BX = 1: the products run on the bf16 matrix pipe as six products of the exact three-way operand split (gemm_bx.hpp): W_hh
// arrives pre-split (k_bx_pack: three planes in fragment order), the state fragment is split by the matrix wave itself.
template <int VARIANT, int TPW, int MW, int BX>
__global__ void __launch_bounds__(256 + 64 * MW) k_gru_chain_fwd(ChainArgs a, const float* __restrict__ gi, float* __restrict__ H,
                                                                  float* __restrict__ saved) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PASSES = CH_SLOTS / MW;
  const int D = a.D, D4 = D >> 2;
  const int G = (VARIANT == TEMP_GRU_TORCH) ? 3 * D : D;
  const ChainGeom g = chain_geom(D);
  const int NT = g.NT, NQ = g.NQ, lda = g.lda, ldh = g.ldh;
  float* accb = lds;                                   // [32][lda]  raw products of the current position
  float* hb = accb + CH_SLOTS * lda;                   // [2][32][ldh] states (double buffered)
  int* tabb = (int*)(hb + 2 * CH_SLOTS * ldh);         // [ms][32] the panel's row table
  float* decb = (float*)(tabb + CH_SLOTS * a.max_steps);   // [ms][32] decay factor of every row
  int* flagb = (int*)(decb + CH_SLOTS * a.max_steps);  // [ms] step flags
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t plane = a.plane;

  for (int p = blockIdx.x; p < a.n_panels; p += gridDim.x) {
    const int rnn_id = a.panel[4 * p], s0 = a.panel[4 * p + 1], ns = a.panel[4 * p + 2];
    const ChainRnn R = a.rnn[rnn_id];
    for (int i = tid; i < 2 * CH_SLOTS * ldh; i += blockDim.x) hb[i] = 0.f;      // k padding of the state rows must be 0, not NaN
    for (int i = tid; i < ns * CH_SLOTS; i += blockDim.x) {
      const int e = a.rows[(size_t)s0 * CH_SLOTS + i];
      tabb[i] = e;
      decb[i] = e >= 0 ? expf(-a.dt[e & CH_ROW_MASK] * a.lambda) : 0.f;
    }
    if (tid < ns) flagb[tid] = a.sinfo[4 * (size_t)(s0 + tid)];
    __syncthreads();

    if (wave < 4) {
      // ------------------------------------------------------------------ matrix role
      const int li = lane & 31, hh = lane >> 5;
      bool tval[TPW];
      int tidx[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) { tidx[j] = wave + 4 * j; tval[j] = tidx[j] < NT; if (!tval[j]) tidx[j] = NT - 1; }
      f32x16 acc[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      if constexpr (BX) {
        // W_hh planes of ONE slab (16 k) in registers: [tile][lane] per plane, packed[((slab * NT + tile) * 3 + plane) * 64 + lane].
        // The products of a slab go plane by plane -- L.ah | M.am, M.ah | H.al, H.am, H.ah (small terms first within a plane),
        // interleaved over the wave's tiles -- so that a plane's registers are free after its last round and are refilled
        // with the NEXT slab's plane at once: 3 to 5 rounds (TPW MFMAs each) of L2 latency cover with a single buffer.
        const bx_u32x4* wp = reinterpret_cast<const bx_u32x4*>(R.wf);
        const int NS = NQ >> 1;
        bx_u32x4 wh[TPW] = {}, wm[TPW] = {}, wl[TPW] = {};
        // (a wave's tile slot past the last tile keeps whatever its registers hold: its products are never stored, and its
        //  share of the plane stream -- 1/20 of the forward's, 1/8 of the backward's -- stays in L2)
        const bool dup_loads = (a.dbg & 128) != 0;
        auto wload = [&](bx_u32x4 (&w)[TPW], int sl, int pl) {
#pragma unroll
          for (int j = 0; j < TPW; ++j)
            if (tval[j] || dup_loads) w[j] = wp[((size_t)(sl * NT + tidx[j]) * 3 + pl) * 64 + lane];
        };
        // Every CU of an XCD walks the same 720 KB of planes; started at the same slab they ask the same L2 lines at the same
        // time.  The walk of a block starts at its own slab (k order of a sum is free; fixed per block, so results stay
        // bit-repeatable).
        const int rot = (a.dbg & 64) ? 0 : (int)(blockIdx.x >> 3) % NS;
        wload(wh, rot, 0); wload(wm, rot, 1); wload(wl, rot, 2);
        for (int s = 0; s < ns; ++s) {
          const int cur = s & 1;
          const int flags = flagb[s];
          if (flags & 1) {
            const int e = tabb[s * CH_SLOTS + li];
            const float decm = (e >= 0 && (e & CH_HAS_PREV)) ? decb[s * CH_SLOTS + li] : 0.f;
            const float* hrow = hb + (size_t)cur * CH_SLOTS * ldh + li * ldh + 8 * hh;     // k = 16 slab + 8 hh .. +7 of track li
            bx_u32x4 AH, AM, AL, NH, NM, NL;
            {
              const float4 r0 = scale4(ld4(hrow + 16 * rot), decm), r1 = scale4(ld4(hrow + 16 * rot + 4), decm);
              unsigned h_, m_, l_;
              bx_split_pair(r0.x, r0.y, h_, m_, l_); AH[0] = h_; AM[0] = m_; AL[0] = l_;
              bx_split_pair(r0.z, r0.w, h_, m_, l_); AH[1] = h_; AM[1] = m_; AL[1] = l_;
              bx_split_pair(r1.x, r1.y, h_, m_, l_); AH[2] = h_; AM[2] = m_; AL[2] = l_;
              bx_split_pair(r1.z, r1.w, h_, m_, l_); AH[3] = h_; AM[3] = m_; AL[3] = l_;
            }
            for (int j = 0, sl = rot; j < NS; ++j) {
              const int sn = sl + 1 < NS ? sl + 1 : 0;            // (after the last slab of the walk: the first one of the NEXT position, weights only)
              const float4 n0 = scale4(ld4(hrow + 16 * sn), decm), n1 = scale4(ld4(hrow + 16 * sn + 4), decm);
              const bx_bf16x8 ah = bx_frag(AH), am = bx_frag(AM), al = bx_frag(AL);
              unsigned h_, m_, l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wl[j]), ah, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              wload(wl, sn, 2);
              bx_split_pair(n0.x, n0.y, h_, m_, l_); NH[0] = h_; NM[0] = m_; NL[0] = l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wm[j]), am, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              bx_split_pair(n0.z, n0.w, h_, m_, l_); NH[1] = h_; NM[1] = m_; NL[1] = l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wm[j]), ah, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              wload(wm, sn, 1);
              bx_split_pair(n1.x, n1.y, h_, m_, l_); NH[2] = h_; NM[2] = m_; NL[2] = l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), al, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              bx_split_pair(n1.z, n1.w, h_, m_, l_); NH[3] = h_; NM[3] = m_; NL[3] = l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), am, acc[j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), ah, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              wload(wh, sn, 0);
              AH = NH; AM = NM; AL = NL;
              sl = sn;
            }
          // lane (li, hh) owns track li and, per register quad qq, gate columns tile*32 + 8qq + 4hh .. +3
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            if (!tval[j]) continue;
            float* dst = accb + (size_t)li * lda + tidx[j] * 32 + 4 * hh;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              st4(dst + 8 * qq, make_float4(acc[j][4 * qq], acc[j][4 * qq + 1], acc[j][4 * qq + 2], acc[j][4 * qq + 3]));
              acc[j][4 * qq] = 0.f; acc[j][4 * qq + 1] = 0.f; acc[j][4 * qq + 2] = 0.f; acc[j][4 * qq + 3] = 0.f;
            }
          }
          }
          __syncthreads();      // A: products of position s are in LDS
          __syncthreads();      // B: states of position s are in LDS
        }
      } else {
      float4 wA[TPW], wB[TPW];
      auto wload = [&](float4 (&w)[TPW], int q) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) w[j] = R.wf[((size_t)tidx[j] * NQ + q) * 64 + lane];
      };
      wload(wA, 0);
      for (int s = 0; s < ns; ++s) {
        const int cur = s & 1;
        const int flags = flagb[s];
        if (flags & 1) {
          const int e = tabb[s * CH_SLOTS + li];
          // decayed previous state (models/RRGCN.py:83); a track without one multiplies (finite) stale LDS contents by 0
          const float decm = (e >= 0 && (e & CH_HAS_PREV)) ? decb[s * CH_SLOTS + li] : 0.f;
          const float* hrow = hb + (size_t)cur * CH_SLOTS * ldh + li * ldh + 4 * hh;
          auto stage = [&](const float4 (&w)[TPW], const float4 h4) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].x, h4.x, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].y, h4.y, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].z, h4.z, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].w, h4.w, acc[j], 0, 0, 0);
          };
          // both operands of stage q + 1 (weights: L2, state: LDS) are requested before the MFMAs of stage q issue; the
          // scheduling barriers keep the compiler from sinking a prefetch down to its first use
          float4 hA = scale4(ld4(hrow), decm), hB;
          for (int q = 0; q < NQ; q += 2) {
            const int q2 = q + 2 < NQ ? q + 2 : 0;
            wload(wB, q + 1);
            hB = scale4(ld4(hrow + 8 * (q + 1)), decm);
            __builtin_amdgcn_sched_barrier(0);
            stage(wA, hA);
            __builtin_amdgcn_sched_barrier(0);
            wload(wA, q2);                                            // past the end: stage 0 of the NEXT position
            hA = scale4(ld4(hrow + 8 * q2), decm);
            __builtin_amdgcn_sched_barrier(0);
            stage(wB, hB);
            __builtin_amdgcn_sched_barrier(0);
          }
          // lane (li, hh) owns track li and, per register quad qq, gate columns tile*32 + 8qq + 4hh .. +3
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            if (!tval[j]) continue;
            float* dst = accb + (size_t)li * lda + tidx[j] * 32 + 4 * hh;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              st4(dst + 8 * qq, make_float4(acc[j][4 * qq], acc[j][4 * qq + 1], acc[j][4 * qq + 2], acc[j][4 * qq + 3]));
              acc[j][4 * qq] = 0.f; acc[j][4 * qq + 1] = 0.f; acc[j][4 * qq + 2] = 0.f; acc[j][4 * qq + 3] = 0.f;
            }
          }
        }
        __syncthreads();      // A: products of position s are in LDS
        __syncthreads();      // B: states of position s are in LDS
      }
      }
    } else {
      // ------------------------------------------------------------------ memory role
      const int mw = wave - 4, c4 = lane, col = 4 * c4;
      const bool cact = c4 < D4;
      float4 bhr = zero4(), bhz = zero4(), bhn = zero4();
      if (cact) { bhr = ld4(R.b_hh + col); bhz = ld4(R.b_hh + D + col); bhn = ld4(R.b_hh + 2 * D + col); }
      int erow[PASSES];
      float4 g0[PASSES], g1[PASSES], g2[PASSES];
      auto prefetch = [&](int s) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int e = tabb[s * CH_SLOTS + ps * MW + mw];
          erow[ps] = e;
          const bool ok = e >= 0 && cact;
          const int er = e & CH_ROW_MASK;
          const float* src = gi + (ok ? (size_t)(a.gi_index ? a.gi_index[er] : er) * G + col : 0);
          if (VARIANT == TEMP_GRU_TORCH) { g0[ps] = ld4(src); g1[ps] = ld4(src + (ok ? D : 0)); g2[ps] = ld4(src + (ok ? 2 * D : 0)); }
          else { g0[ps] = zero4(); g1[ps] = zero4(); g2[ps] = ld4(src); }
        }
      };
      prefetch(0);
      for (int s = 0; s < ns; ++s) {
        const int cur = s & 1;
        const int flags = flagb[s];
        __syncthreads();      // A
        const float* hcur = hb + (size_t)cur * CH_SLOTS * ldh;
        float* hnext = hb + (size_t)(cur ^ 1) * CH_SLOTS * ldh;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int slot = ps * MW + mw;
          const int e = erow[ps];
          if (e < 0 || !cact) continue;
          const size_t row = (size_t)(e & CH_ROW_MASK);
          const bool hp = (e & CH_HAS_PREV) != 0;
          float4 ar = zero4(), az = zero4(), an = zero4(), hd = zero4();
          if (hp) {                                   // (a track without a previous state contributed zeros to the products)
            const float* ab = accb + (size_t)slot * lda + col;
            ar = ld4(ab); az = ld4(ab + D); an = ld4(ab + 2 * D);
            hd = scale4(ld4(hcur + (size_t)slot * ldh + col), decb[s * CH_SLOTS + slot]);
          }
          float o_h[4], o_r[4], o_z[4], o_n[4], o_hn[4];
          const float arv[4] = {ar.x, ar.y, ar.z, ar.w}, azv[4] = {az.x, az.y, az.z, az.w}, anv[4] = {an.x, an.y, an.z, an.w};
          const float hdv[4] = {hd.x, hd.y, hd.z, hd.w};
          const float g0v[4] = {g0[ps].x, g0[ps].y, g0[ps].z, g0[ps].w}, g1v[4] = {g1[ps].x, g1[ps].y, g1[ps].z, g1[ps].w};
          const float g2v[4] = {g2[ps].x, g2[ps].y, g2[ps].z, g2[ps].w};
          const float brv[4] = {bhr.x, bhr.y, bhr.z, bhr.w}, bzv[4] = {bhz.x, bhz.y, bhz.z, bhz.w}, bnv[4] = {bhn.x, bhn.y, bhn.z, bhn.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float xr = arv[k], xz = azv[k];
            if (VARIANT == TEMP_GRU_TORCH) { xr += g0v[k]; xz += g1v[k]; }
            const float rg = gate_sigmoid(xr + brv[k]);
            const float zg = gate_sigmoid(xz + bzv[k]);
            const float hn = anv[k] + bnv[k];
            const float ng = gate_tanh(g2v[k] + rg * hn);
            o_h[k] = (VARIANT == TEMP_GRU_TORCH) ? ((1.f - zg) * ng + zg * hdv[k]) : (ng + zg * (hdv[k] - ng));
            o_r[k] = rg; o_z[k] = zg; o_n[k] = ng; o_hn[k] = hn;
          }
          const float4 h4 = make_float4(o_h[0], o_h[1], o_h[2], o_h[3]);
          st4(hnext + (size_t)slot * ldh + col, h4);
          const size_t o = row * D + col;
          if (flags & 2) st4(H + o, h4);
          st4(saved + o, make_float4(o_r[0], o_r[1], o_r[2], o_r[3]));
          st4(saved + plane + o, make_float4(o_z[0], o_z[1], o_z[2], o_z[3]));
          st4(saved + 2 * plane + o, make_float4(o_n[0], o_n[1], o_n[2], o_n[3]));
          st4(saved + 3 * plane + o, make_float4(o_hn[0], o_hn[1], o_hn[2], o_hn[3]));
          st4(saved + 4 * plane + o, hd);
        }
        __syncthreads();      // B
        // issued AFTER the barrier: the 24 loads + table look-ups of a prefetch take ~4 k cycles to issue, which the matrix waves
        // would otherwise spend waiting at B (measured with s_memtime stamps: 9 % of a position)
        if (s + 1 < ns) prefetch(s + 1);       // in flight while the matrix waves run position s + 1
      }
    }
    __syncthreads();          // LDS is re-initialised for the next panel
  }
}

// ---- backward -------------------------------------------------------------------------------------------------------
// G4 = 1 (nn.GRU gate layout): the gate gradients are written ONCE, as dgi = [n][4d] = [dr | dz | dn_i | dn_h] (dgh unused) -- two
// thirds of dgh repeat dgi; the weight-gradient and d_x kernels address their columns of the one matrix (gru_wgrad.hpp).
template <int VARIANT, int TPWB, int MW, int BX, int G4 = 0>
__global__ void __launch_bounds__(256 + 64 * MW) k_gru_chain_bwd(ChainArgs a, ChainUps ups, const float* __restrict__ saved,
                                                                  float* __restrict__ dgi, float* __restrict__ dgh) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PASSES = CH_SLOTS / MW;
  const int D = a.D, D4 = D >> 2;
  const int G = (VARIANT == TEMP_GRU_TORCH) ? 3 * D : D;
  const ChainGeom g = chain_geom(D);
  const int NTb = g.NTb, NQb = g.NQb, ldA = g.ldA, ldz = g.ldz;
  float* ab = lds;                                     // [32][ldA]  gate gradients w.r.t. the recurrent pre-activations (dgh)
  float* gzb = ab + CH_SLOTS * ldA;                    // [32][ldz]  dh * z
  float* dpb = gzb + CH_SLOTS * ldz;                   // [32][ldz]  d_prev of the position just processed
  int* tabb = (int*)(dpb + CH_SLOTS * ldz);            // [ms][32] the panel's row table
  float* decb = (float*)(tabb + CH_SLOTS * a.max_steps);   // [ms][32]
  int* flagb = (int*)(decb + CH_SLOTS * a.max_steps);  // [ms]
  int* upb = flagb + a.max_steps;                      // [ms][2] upstream block and its first row of every step
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t plane = a.plane;

  for (int p = blockIdx.x; p < a.n_panels; p += gridDim.x) {
    const int rnn_id = a.panel[4 * p], s0 = a.panel[4 * p + 1], ns = a.panel[4 * p + 2];
    const ChainRnn R = a.rnn[rnn_id];
    for (int i = tid; i < CH_SLOTS * ldA; i += blockDim.x) ab[i] = 0.f;          // k padding of the dgh rows
    if (tid < ns) { upb[2 * tid] = a.sinfo[4 * (size_t)(s0 + tid) + 1]; upb[2 * tid + 1] = a.sinfo[4 * (size_t)(s0 + tid) + 2]; }
    for (int i = tid; i < ns * CH_SLOTS; i += blockDim.x) {
      const int e = a.rows[(size_t)s0 * CH_SLOTS + i];
      tabb[i] = e;
      decb[i] = e >= 0 ? expf(-a.dt[e & CH_ROW_MASK] * a.lambda) : 0.f;
    }
    if (tid < ns) flagb[tid] = a.sinfo[4 * (size_t)(s0 + tid)];
    __syncthreads();

    if (wave < 4) {
      // ------------------------------------------------------------------ matrix role: d_prev = (dgh . W_hh + dh*z) * decay
      const int li = lane & 31, hh = lane >> 5;
      bool tval[TPWB];
      int tidx[TPWB];
#pragma unroll
      for (int j = 0; j < TPWB; ++j) { tidx[j] = wave + 4 * j; tval[j] = tidx[j] < NTb; if (!tval[j]) tidx[j] = NTb - 1; }
      f32x16 acc[TPWB];
#pragma unroll
      for (int j = 0; j < TPWB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      if constexpr (BX) {
        // W_hh planes of TWO slabs (16 k each) in registers; the planes of slab sl + 2 replace those of slab sl as soon as
        // their last round of slab sl has issued (L after round 0, M after round 2, H after round 5): nine to eleven rounds of
        // L2 latency cover.  NQb is a multiple of 4, so the slab count is even.
        const bx_u32x4* wp = reinterpret_cast<const bx_u32x4*>(R.wb);
        const int NS = NQb >> 1;
        bx_u32x4 wh0[TPWB] = {}, wm0[TPWB] = {}, wl0[TPWB] = {}, wh1[TPWB] = {}, wm1[TPWB] = {}, wl1[TPWB] = {};
        const bool dup_loads = (a.dbg & 128) != 0;
        auto wload = [&](bx_u32x4 (&w)[TPWB], int sl, int pl) {
#pragma unroll
          for (int j = 0; j < TPWB; ++j)
            if (tval[j] || dup_loads) w[j] = wp[((size_t)(sl * NTb + tidx[j]) * 3 + pl) * 64 + lane];
        };
        const int rot = (a.dbg & 64) ? 0 : 2 * ((int)(blockIdx.x >> 3) % (NS >> 1));     // per-block start of the slab walk (see the forward kernel)
        wload(wh0, rot, 0); wload(wm0, rot, 1); wload(wl0, rot, 2);
        wload(wh1, rot + 1, 0); wload(wm1, rot + 1, 1); wload(wl1, rot + 1, 2);
        for (int s = ns - 1; s >= 0; --s) {
          const int flags = flagb[s];
          __syncthreads();      // A: dgh / dh*z / decay of position s are in LDS
          if (flags & 1) {
            const float* arow = ab + (size_t)li * ldA + 8 * hh;                 // k = 16 slab + 8 hh .. +7 of track li
            bx_u32x4 AH, AM, AL, NH, NM, NL;
            {
              const float4 r0 = ld4(arow + 16 * rot), r1 = ld4(arow + 16 * rot + 4);
              unsigned h_, m_, l_;
              bx_split_pair(r0.x, r0.y, h_, m_, l_); AH[0] = h_; AM[0] = m_; AL[0] = l_;
              bx_split_pair(r0.z, r0.w, h_, m_, l_); AH[1] = h_; AM[1] = m_; AL[1] = l_;
              bx_split_pair(r1.x, r1.y, h_, m_, l_); AH[2] = h_; AM[2] = m_; AL[2] = l_;
              bx_split_pair(r1.z, r1.w, h_, m_, l_); AH[3] = h_; AM[3] = m_; AL[3] = l_;
            }
            // one slab out of the plane registers (wh, wm, wl); `sl2` = the slab they are refilled with
            auto slab = [&](bx_u32x4 (&wh)[TPWB], bx_u32x4 (&wm)[TPWB], bx_u32x4 (&wl)[TPWB], int sl, int sl2) {
              const int sn = sl + 1 < NS ? sl + 1 : 0;
              const float4 n0 = ld4(arow + 16 * sn), n1 = ld4(arow + 16 * sn + 4);
              const bx_bf16x8 ah = bx_frag(AH), am = bx_frag(AM), al = bx_frag(AL);
              unsigned h_, m_, l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wl[j]), ah, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              wload(wl, sl2, 2);
              bx_split_pair(n0.x, n0.y, h_, m_, l_); NH[0] = h_; NM[0] = m_; NL[0] = l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wm[j]), am, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              bx_split_pair(n0.z, n0.w, h_, m_, l_); NH[1] = h_; NM[1] = m_; NL[1] = l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wm[j]), ah, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              wload(wm, sl2, 1);
              bx_split_pair(n1.x, n1.y, h_, m_, l_); NH[2] = h_; NM[2] = m_; NL[2] = l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), al, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              bx_split_pair(n1.z, n1.w, h_, m_, l_); NH[3] = h_; NM[3] = m_; NL[3] = l_;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), am, acc[j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), ah, acc[j], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              wload(wh, sl2, 0);
              AH = NH; AM = NM; AL = NL;
            };
            for (int j = 0, sl = rot; j < NS; j += 2) {
              const int a2 = sl + 2 < NS ? sl + 2 : 0, b2 = a2 + 1;          // (after the walk's last pair: the first pair of the NEXT position)
              slab(wh0, wm0, wl0, sl, a2);
              slab(wh1, wm1, wl1, sl + 1, b2);
              sl = a2;
            }
          const float dec = decb[s * CH_SLOTS + li];
#pragma unroll
          for (int j = 0; j < TPWB; ++j) {
            if (!tval[j]) continue;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const int c = tidx[j] * 32 + 8 * qq + 4 * hh;
              const float4 gz = ld4(gzb + (size_t)li * ldz + c);
              st4(dpb + (size_t)li * ldz + c, make_float4((acc[j][4 * qq] + gz.x) * dec, (acc[j][4 * qq + 1] + gz.y) * dec,
                                                           (acc[j][4 * qq + 2] + gz.z) * dec, (acc[j][4 * qq + 3] + gz.w) * dec));
              acc[j][4 * qq] = 0.f; acc[j][4 * qq + 1] = 0.f; acc[j][4 * qq + 2] = 0.f; acc[j][4 * qq + 3] = 0.f;
            }
          }
          }
          __syncthreads();      // B: d_prev of position s is in LDS
        }
      } else {
      // weights: a ring of four stage buffers, i.e. three stages (3 x 8 MFMAs = 1.5 k cycles) of L2 latency cover
      float4 w0[TPWB], w1[TPWB], w2[TPWB], w3[TPWB];
      auto wload = [&](float4 (&w)[TPWB], int q) {
#pragma unroll
        for (int j = 0; j < TPWB; ++j) w[j] = R.wb[((size_t)tidx[j] * NQb + q) * 64 + lane];
      };
      wload(w0, 0); wload(w1, 1); wload(w2, 2);
      for (int s = ns - 1; s >= 0; --s) {
        const int cur = s & 1;
        const int flags = flagb[s];
        __syncthreads();      // A: dgh / dh*z / decay of position s are in LDS
        if (flags & 1) {
          const float* arow = ab + (size_t)li * ldA + 4 * hh;
          auto stage = [&](const float4 (&w)[TPWB], const float4 d4) {
#pragma unroll
            for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].x, d4.x, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].y, d4.y, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].z, d4.z, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].w, d4.w, acc[j], 0, 0, 0);
          };
          float4 dA = ld4(arow), dB;
          for (int q = 0; q < NQb; q += 4) {
            const int qa = q + 4 < NQb ? q + 4 : 0, qb = q + 5 < NQb ? q + 5 : 1, qc = q + 6 < NQb ? q + 6 : 2;   // wrap: the NEXT position
            wload(w3, q + 3);
            dB = ld4(arow + 8 * (q + 1));
            __builtin_amdgcn_sched_barrier(0);
            stage(w0, dA);
            __builtin_amdgcn_sched_barrier(0);
            wload(w0, qa);
            dA = ld4(arow + 8 * (q + 2));
            __builtin_amdgcn_sched_barrier(0);
            stage(w1, dB);
            __builtin_amdgcn_sched_barrier(0);
            wload(w1, qb);
            dB = ld4(arow + 8 * (q + 3));
            __builtin_amdgcn_sched_barrier(0);
            stage(w2, dA);
            __builtin_amdgcn_sched_barrier(0);
            wload(w2, qc);
            dA = ld4(arow + 8 * (q + 4 < NQb ? q + 4 : 0));
            __builtin_amdgcn_sched_barrier(0);
            stage(w3, dB);
            __builtin_amdgcn_sched_barrier(0);
          }
          const float dec = decb[s * CH_SLOTS + li];
#pragma unroll
          for (int j = 0; j < TPWB; ++j) {
            if (!tval[j]) continue;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const int c = tidx[j] * 32 + 8 * qq + 4 * hh;
              const float4 gz = ld4(gzb + (size_t)li * ldz + c);
              st4(dpb + (size_t)li * ldz + c, make_float4((acc[j][4 * qq] + gz.x) * dec, (acc[j][4 * qq + 1] + gz.y) * dec,
                                                           (acc[j][4 * qq + 2] + gz.z) * dec, (acc[j][4 * qq + 3] + gz.w) * dec));
              acc[j][4 * qq] = 0.f; acc[j][4 * qq + 1] = 0.f; acc[j][4 * qq + 2] = 0.f; acc[j][4 * qq + 3] = 0.f;
            }
          }
        }
        __syncthreads();      // B: d_prev of position s is in LDS
      }
      }
    } else {
      // ------------------------------------------------------------------ memory role: gate gradients
      const int mw = wave - 4, c4 = lane, col = 4 * c4;
      const bool cact = c4 < D4;
      int erow[PASSES];
      bool nxt[PASSES];
      float4 sr[PASSES], sz[PASSES], sn[PASSES], shn[PASSES], shd[PASSES];
      auto prefetch = [&](int s) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int slot = ps * MW + mw;
          const int e = tabb[s * CH_SLOTS + slot];
          erow[ps] = e;
          int en = -1;
          if (s + 1 < ns) en = tabb[(s + 1) * CH_SLOTS + slot];
          nxt[ps] = en >= 0 && (en & CH_HAS_PREV);
          const bool ok = e >= 0 && cact;
          const size_t row = ok ? (size_t)(e & CH_ROW_MASK) : 0;
          const float* src = saved + row * D + (ok ? col : 0);
          sr[ps] = ld4(src); sz[ps] = ld4(src + plane); sn[ps] = ld4(src + 2 * plane); shn[ps] = ld4(src + 3 * plane);
          shd[ps] = ld4(src + 4 * plane);
        }
      };
      prefetch(ns - 1);
      for (int s = ns - 1; s >= 0; --s) {
        const int cur = s & 1;
        // upstream gradient of the step's rows (only the positions whose states are consumed outside the chain -- the
        // target, the last history position -- have one: loaded on demand instead of holding registers for it all the time)
        const int up_sel = upb[2 * s], up_row0 = upb[2 * s + 1];       // (staged in LDS: two dependent global loads per step otherwise)
        const float* upp = up_sel >= 0 ? ups.p[up_sel] : nullptr;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int slot = ps * MW + mw;
          const int e = erow[ps];
          if (!cact) continue;
          if (e < 0) continue;                 // idle track: whatever its LDS rows hold only reaches its own, unread, d_prev row
          const size_t row = (size_t)(e & CH_ROW_MASK);
          float4 gd = upp ? ld4(upp + (row - (size_t)up_row0) * D + col) : zero4();
          if (nxt[ps]) gd = add4(gd, ld4(dpb + (size_t)slot * ldz + col));
          const float4 rg = sr[ps], zg = sz[ps], ng = sn[ps], hn = shn[ps], hd = shd[ps];
          float4 dr_pre, dz_pre, dn_pre, dhn, gz;
#define TEMP_GATE(c)                                          \
          {                                                   \
            const float dn = gd.c * (1.f - zg.c);             \
            const float dz = gd.c * (hd.c - ng.c);            \
            dn_pre.c = dn * (1.f - ng.c * ng.c);              \
            dr_pre.c = dn_pre.c * hn.c * rg.c * (1.f - rg.c); \
            dz_pre.c = dz * zg.c * (1.f - zg.c);              \
            dhn.c = dn_pre.c * rg.c;                          \
            gz.c = gd.c * zg.c;                               \
          }
          TEMP_GATE(x) TEMP_GATE(y) TEMP_GATE(z) TEMP_GATE(w)
#undef TEMP_GATE
          float* arow = ab + (size_t)slot * ldA + col;
          st4(arow, dr_pre); st4(arow + D, dz_pre); st4(arow + 2 * D, dhn);
          st4(gzb + (size_t)slot * ldz + col, gz);
          if constexpr (G4) {
            const size_t b4 = row * 4 * D + col;
            st4(dgi + b4, dr_pre); st4(dgi + b4 + D, dz_pre); st4(dgi + b4 + 2 * D, dn_pre); st4(dgi + b4 + 3 * D, dhn);
          } else {
          const size_t b3 = row * 3 * D + col;
          if (VARIANT == TEMP_GRU_TORCH) { st4(dgi + b3, dr_pre); st4(dgi + b3 + D, dz_pre); st4(dgi + b3 + 2 * D, dn_pre); }
          else st4(dgi + row * D + col, dn_pre);
          st4(dgh + b3, dr_pre); st4(dgh + b3 + D, dz_pre); st4(dgh + b3 + 2 * D, dhn);
          }
        }
        __syncthreads();      // A
        if (s > 0) prefetch(s - 1);            // issued behind the barrier (the matrix waves start at once), in flight while they
        __syncthreads();      // B             // run position s
      }
    }
    __syncthreads();
  }
}

static int chain_check(const TempGruChain* c) {
  if (!c || c->d <= 0 || c->n_panels < 0 || c->n_steps < 0 || c->n_rnn <= 0 || c->n_rnn > TEMP_CHAIN_MAX_RNN) return TEMP_E_BADARG;
  if (c->variant != TEMP_GRU_TORCH && c->variant != TEMP_GRU_TYPE1) return TEMP_E_BADARG;
  if (c->n_panels > 0 && (!c->panel || !c->rows || !c->sinfo || !c->dt)) return TEMP_E_BADARG;
  for (int i = 0; i < c->n_rnn; ++i) if (!c->packed[i] || !c->b_hh[i]) return TEMP_E_BADARG;
  if (c->max_steps <= 0 || c->max_steps > CH_MAX_STEPS) return TEMP_E_BADARG;
  if (!temp_gru_chain_supported(c->d)) return TEMP_E_UNSUPPORTED;
  return TEMP_OK;
}

// the products of the chain run on the bf16 matrix pipe (three-way split) unless TEMP_MFMA=f32 or d is not a multiple of 8
static bool chain_bx(int d) { return bx_enabled() && d % 8 == 0; }
// ... and on the f16 pipe as three products of the scaled two-way split (gru_chain_hx.hpp) unless TEMP_MFMA=bf16x3 or its LDS images do not fit
static bool chain_hx(int d) {
  if (!chain_bx(d) || !hx_enabled()) return false;
  const ChainGeomHx g = chain_geom_hx(d);
  return chain_lds_fwd_hx(d, CH_MAX_STEPS) <= CH_LDS_LIMIT && chain_lds_bwd_hx(d, CH_MAX_STEPS) <= CH_LDS_LIMIT && g.NT <= 20 && g.NTb <= 8;   // (six tiles per matrix wave: the two register sets spill)
}

static ChainArgs chain_args(const TempGruChain* c) {
  ChainArgs a = {};
  const ChainGeom g = chain_geom(c->d);
  a.D = c->d; a.n_panels = c->n_panels; a.max_steps = c->max_steps; a.panel = c->panel; a.rows = c->rows; a.sinfo = c->sinfo; a.dt = c->dt;
  a.n_rnn_keys = c->n_rnn; a.lambda = c->lambda; a.plane = c->saved_plane; a.gi_index = c->gi_index; a.dbg = option(TEMP_OPT_DEBUG) >> 8;      // development A/B switches (bit 6: no per-block rotation of the slab walk); 0 in every product run
  for (int i = 0; i < c->n_rnn; ++i) {
    a.rnn[i].wf = (const float4*)c->packed[i];
    a.rnn[i].wb = (const float4*)c->packed[i] + (chain_bx(c->d) ? (size_t)(g.NQ >> 1) * g.NT * 192 : (size_t)g.NT * g.NQ * 64);
    a.rnn[i].b_hh = c->b_hh[i];
    a.rnn[i].kf = a.rnn[i].kb = nullptr;
    if (chain_hx(c->d)) {
      const ChainGeomHx gx = chain_geom_hx(c->d);
      a.rnn[i].wb = (const float4*)c->packed[i] + chain_hx_fwd_items(c->d);
      a.rnn[i].kf = (const unsigned*)((const float4*)c->packed[i] + chain_hx_fwd_items(c->d) + chain_hx_bwd_items(c->d));
      a.rnn[i].kb = a.rnn[i].kf + gx.NT * 32;
    }
  }
  return a;
}

template <class K>
static int chain_lds_attr(K kernel, size_t bytes, bool* done) {
  if (*done) return TEMP_OK;
  if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_LIMIT) != hipSuccess) return TEMP_E_LAUNCH;
  (void)bytes;
  *done = true;
  return TEMP_OK;
}

template <int VARIANT, int TPW>
static int launch_chain_fwd(const ChainArgs& a, const float* gi, float* h, float* saved, hipStream_t st) {
  static bool attr = false;
  static bool attr_bx = false;
  static bool attr_hx = false;
  if (chain_hx(a.D)) {
    const size_t lds_hx = chain_lds_fwd_hx(a.D, a.max_steps);
    // 8 + 8 waves (two matrix and two memory waves per SIMD, 128 registers each): a memory wave then has 28 stores + 12 loads
    // in flight per position instead of 56 + 24 -- beyond the 63 a wave's counter can track, every further access waits for the oldest
    constexpr int TPW8 = (TPW * 4 + 7) / 8 < 1 ? 1 : (TPW * 4 + 7) / 8;
    const int cfg = (a.dbg >> 4) & 3;                          // development A/B: 0 = 8 + 8 waves, 1 = 4 + 4, 2 = 4 + 8
    if (cfg == 1) {
      auto kernel = k_gru_chain_fwd_hx<VARIANT, TPW, 4, 4>;
      int rc = chain_lds_attr(kernel, lds_hx, &attr_hx);
      if (rc) return rc;
      TEMP_LAUNCH(K_GRU_CHAIN_FWD, kernel, dim3(a.n_panels), dim3(512), lds_hx, st, a, gi, h, saved);
    } else if (cfg == 2) {
      static bool attr_48 = false;
      auto kernel = k_gru_chain_fwd_hx<VARIANT, TPW, 8, 4>;
      int rc = chain_lds_attr(kernel, lds_hx, &attr_48);
      if (rc) return rc;
      TEMP_LAUNCH(K_GRU_CHAIN_FWD, kernel, dim3(a.n_panels), dim3(768), lds_hx, st, a, gi, h, saved);
    } else if (cfg == 3) {                                      // (development A/B: 8 + 8 waves, TWO register sets of planes: 37 spills, 291 us against 233)
      static bool attr_882 = false;
      auto kernel = k_gru_chain_fwd_hx<VARIANT, TPW8, 8, 8, 1>;
      int rc = chain_lds_attr(kernel, lds_hx, &attr_882);
      if (rc) return rc;
      TEMP_LAUNCH(K_GRU_CHAIN_FWD, kernel, dim3(a.n_panels), dim3(1024), lds_hx, st, a, gi, h, saved);
    } else {
      static bool attr_88 = false;
      auto kernel = k_gru_chain_fwd_hx<VARIANT, TPW8, 8, 8>;
      int rc = chain_lds_attr(kernel, lds_hx, &attr_88);
      if (rc) return rc;
      TEMP_LAUNCH(K_GRU_CHAIN_FWD, kernel, dim3(a.n_panels), dim3(1024), lds_hx, st, a, gi, h, saved);
    }
    hx_count();
    return launch_status();
  }
  const size_t lds = chain_lds_fwd(a.D, a.max_steps);
  if (chain_bx(a.D)) {
    auto kernel = k_gru_chain_fwd<VARIANT, TPW, 4, 1>;
    int rc = chain_lds_attr(kernel, lds, &attr_bx);
    if (rc) return rc;
    TEMP_LAUNCH(K_GRU_CHAIN_FWD, kernel, dim3(a.n_panels), dim3(512), lds, st, a, gi, h, saved);
    return launch_status();
  }
  auto kernel = k_gru_chain_fwd<VARIANT, TPW, 4, 0>;
  int rc = chain_lds_attr(kernel, lds, &attr);
  if (rc) return rc;
  TEMP_LAUNCH(K_GRU_CHAIN_FWD, kernel, dim3(a.n_panels), dim3(512), lds, st, a, gi, h, saved);
  return launch_status();
}

template <int VARIANT, int TPWB, int G4 = 0>
static int launch_chain_bwd(const ChainArgs& a, const ChainUps& ups, const float* saved, float* dgi, float* dgh, hipStream_t st,
                            unsigned* row_keys = nullptr, unsigned* col_keys = nullptr) {
  static bool attr = false;
  static bool attr_bx = false;
  static bool attr_hx = false;
  if (chain_hx(a.D)) {
    const size_t lds_hx = chain_lds_bwd_hx(a.D, a.max_steps);
    constexpr int TPWB8 = (TPWB * 4 + 7) / 8 < 1 ? 1 : (TPWB * 4 + 7) / 8;
    // 4 + 8 waves (168 registers); 8 + 8 waves with a ring of eight slabs -- two matrix waves per SIMD covering each other's L2
    // latency -- leaves the memory role 128 registers: 31 spills, 374 us against 306 (development A/B: TEMP_DEBUG = 8192)
    const int cfg = (a.dbg >> 4) & 3;
    if (cfg == 3) {                                            // (development A/B: a ring of five slabs)
      static bool attr_5 = false;
      auto kernel = k_gru_chain_bwd_hx<VARIANT, TPWB, 8, G4, 4, 5>;
      int rc = chain_lds_attr(kernel, lds_hx, &attr_5);
      if (rc) return rc;
      TEMP_LAUNCH(K_GRU_CHAIN_BWD, kernel, dim3(a.n_panels), dim3(768), lds_hx, st, a, ups, saved, dgi, dgh, row_keys, col_keys);
    } else if (cfg != 2) {
      auto kernel = k_gru_chain_bwd_hx<VARIANT, TPWB, 8, G4, 4, 4>;
      int rc = chain_lds_attr(kernel, lds_hx, &attr_hx);
      if (rc) return rc;
      TEMP_LAUNCH(K_GRU_CHAIN_BWD, kernel, dim3(a.n_panels), dim3(768), lds_hx, st, a, ups, saved, dgi, dgh, row_keys, col_keys);
    } else {
      static bool attr_88 = false;
      auto kernel = k_gru_chain_bwd_hx<VARIANT, TPWB8, 8, G4, 8, 8>;
      int rc = chain_lds_attr(kernel, lds_hx, &attr_88);
      if (rc) return rc;
      TEMP_LAUNCH(K_GRU_CHAIN_BWD, kernel, dim3(a.n_panels), dim3(1024), lds_hx, st, a, ups, saved, dgi, dgh, row_keys, col_keys);
    }
    hx_count();
    if (col_keys) TEMP_LAUNCH(K_GRU_CHAIN_PACK, k_keys_reduce, dim3(ceil_div(4 * a.D, 32), a.n_rnn_keys), dim3(1024), 0, st, a.n_panels, 4 * a.D, col_keys + (size_t)a.n_rnn_keys * 4 * a.D, col_keys, a.panel, 4);
    return launch_status();
  }
  if (row_keys || col_keys) return TEMP_E_UNSUPPORTED;          // (only the f16 kernels produce keys: ask temp_gru_chain_keys_supported first)
  const size_t lds = chain_lds_bwd(a.D, a.max_steps);
  if (chain_bx(a.D)) {
    auto kernel = k_gru_chain_bwd<VARIANT, TPWB, 8, 1, G4>;
    int rc = chain_lds_attr(kernel, lds, &attr_bx);
    if (rc) return rc;
    TEMP_LAUNCH(K_GRU_CHAIN_BWD, kernel, dim3(a.n_panels), dim3(768), lds, st, a, ups, saved, dgi, dgh);
    return launch_status();
  }
  auto kernel = k_gru_chain_bwd<VARIANT, TPWB, 8, 0, G4>;
  int rc = chain_lds_attr(kernel, lds, &attr);
  if (rc) return rc;
  TEMP_LAUNCH(K_GRU_CHAIN_BWD, kernel, dim3(a.n_panels), dim3(768), lds, st, a, ups, saved, dgi, dgh);
  return launch_status();
}

}  // namespace temp

using namespace temp;

extern "C" {

int temp_gru_chain_supported(int d) {
  if (d <= 0 || d % 4) return 0;
  return chain_lds_fwd(d, CH_MAX_STEPS) <= CH_LDS_LIMIT && chain_lds_bwd(d, CH_MAX_STEPS) <= CH_LDS_LIMIT && chain_geom(d).NT <= 24 && chain_geom(d).NTb <= 8;
}

size_t temp_gru_chain_pack_floats(int d) {
  if (d <= 0) return 0;
  const ChainGeom g = chain_geom(d);
  const size_t f32 = ((size_t)g.NT * g.NQ + (size_t)g.NTb * g.NQb) * 64 * 4;
  // three bf16 planes in fragment order: (slabs of 16 k) x tiles x 192 sixteen-byte items, forward then backward
  const size_t bx = ((size_t)(g.NQ >> 1) * g.NT + (size_t)(g.NQb >> 1) * g.NTb) * 192 * 4;
  const size_t hx = chain_hx_pack_floats(d);           // two f16 planes + column keys (gru_chain_hx.hpp)
  const size_t m = f32 > bx ? f32 : bx;
  return m > hx ? m : hx;                              // any arithmetic (TEMP_MFMA) fits the caller's buffer
}

int temp_gru_chain_pack(int d, const float* w_hh, float* packed, void* stream) {
  if (d <= 0 || !w_hh || !packed) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  const ChainGeom g = chain_geom(d);
  if (chain_hx(d)) return temp_gru_chain_pack_multi(1, d, &w_hh, &packed, stream);
  if (chain_bx(d)) {
    // forward: gate column x k = W_hh as stored ([3d][d], k contiguous); backward: k = gate column, state column = W_hh as [K][N]
    const int nsf = g.NQ >> 1, nsb = g.NQb >> 1;
    bx_u32x4* pf = reinterpret_cast<bx_u32x4*>(packed);
    bx_u32x4* pb = pf + (size_t)nsf * g.NT * 192;
    TEMP_LAUNCH(K_GRU_CHAIN_PACK, (k_bx_pack<1>), dim3(ceil_div((long long)nsf * g.NT, 4)), dim3(256), 0, (hipStream_t)stream, d, 3 * d, g.NT, nsf,
                w_hh, d, pf);
    TEMP_LAUNCH(K_GRU_CHAIN_PACK, (k_bx_pack<0>), dim3(ceil_div((long long)nsb * g.NTb, 4)), dim3(256), 0, (hipStream_t)stream, 3 * d, d, g.NTb, nsb,
                w_hh, d, pb);
    return launch_status();
  }
  const size_t n4 = ((size_t)g.NT * g.NQ + (size_t)g.NTb * g.NQb) * 64;
  int gx = ceil_div((long long)n4, 256);
  if (gx > 1024) gx = 1024;
  TEMP_LAUNCH(K_GRU_CHAIN_PACK, k_gru_chain_pack, dim3(gx), dim3(256), 0, (hipStream_t)stream, d, w_hh, (float4*)packed);
  return launch_status();
}

int temp_gru_chain_pack_multi(int count, int d, const float* const* w_hh, float* const* packed, void* stream) {
  if (count <= 0 || count > TEMP_CHAIN_MAX_RNN || d <= 0 || !w_hh || !packed) return TEMP_E_BADARG;
  for (int i = 0; i < count; ++i) if (!w_hh[i] || !packed[i]) return TEMP_E_BADARG;
  if (d % 4) return TEMP_E_UNSUPPORTED;
  if (chain_hx(d)) {
    // two f16 planes with per-column scales: forward (gate column x k: W_hh as stored), backward (k = gate column, state column:
    // W_hh as [K][N], its slab count rounded up to the kernel's multiple of four), then the keys of both
    const ChainGeomHx gx = chain_geom_hx(d);
    HxPackJobs jobs = {};
    for (int i = 0; i < count; ++i) {
      hx_u32x4* pf = reinterpret_cast<hx_u32x4*>(packed[i]);
      hx_u32x4* pb = pf + chain_hx_fwd_items(d);
      unsigned* kf = reinterpret_cast<unsigned*>(pb + chain_hx_bwd_items(d));
      if (jobs.count + 2 > HX_PACK_JOBS) { hx_pack_launch(jobs, K_GRU_CHAIN_PACK, (hipStream_t)stream); jobs = HxPackJobs{}; }
      hx_pack_jobs_add(jobs, w_hh[i], pf, kf, d, 3 * d, d, 1);
      hx_pack_jobs_add(jobs, w_hh[i], pb, kf + gx.NT * 32, 3 * d, d, d, 0, gx.NSb);
    }
    hx_pack_launch(jobs, K_GRU_CHAIN_PACK, (hipStream_t)stream);
    return launch_status();
  }
  if (!chain_bx(d) || 2 * count > BX_PACK_JOBS) {
    for (int i = 0; i < count; ++i) { const int rc = temp_gru_chain_pack(d, w_hh[i], packed[i], stream); if (rc) return rc; }
    return TEMP_OK;
  }
  const ChainGeom g = chain_geom(d);
  BxPackJobs jobs = {};
  for (int i = 0; i < count; ++i) {                           // forward planes (W_hh as stored: [3d][d], k contiguous), then backward planes (W_hh as [K][N])
    bx_u32x4* pf = reinterpret_cast<bx_u32x4*>(packed[i]);
    bx_pack_jobs_add(jobs, w_hh[i], pf, d, 3 * d, d, 1);
    bx_pack_jobs_add(jobs, w_hh[i], pf + (size_t)(g.NQ >> 1) * g.NT * 192, 3 * d, d, d, 0);
  }
  TEMP_LAUNCH(K_GRU_CHAIN_PACK, k_bx_pack_multi, dim3(ceil_div(jobs.total_units, 4)), dim3(256), 0, (hipStream_t)stream, jobs);
  return launch_status();
}

int temp_gru_chain_fwd(const TempGruChain* c, const float* gi, float* h_out, float* saved, void* stream) {
  int rc = chain_check(c);
  if (rc) return rc;
  if (c->n_panels == 0) return TEMP_OK;
  if (!gi || !h_out || !saved) return TEMP_E_BADARG;
  const ChainArgs a = chain_args(c);
  hipStream_t st = (hipStream_t)stream;
  const int tpw = ceil_div(chain_geom(c->d).NT, 4);
#define TEMP_CHAIN_FWD(V)                                                            \
  switch (tpw) {                                                                     \
    case 1: return launch_chain_fwd<V, 1>(a, gi, h_out, saved, st);                  \
    case 2: return launch_chain_fwd<V, 2>(a, gi, h_out, saved, st);                  \
    case 3: return launch_chain_fwd<V, 3>(a, gi, h_out, saved, st);                  \
    case 4: return launch_chain_fwd<V, 4>(a, gi, h_out, saved, st);                  \
    case 5: return launch_chain_fwd<V, 5>(a, gi, h_out, saved, st);                  \
    case 6: return launch_chain_fwd<V, 6>(a, gi, h_out, saved, st);                  \
    default: return TEMP_E_UNSUPPORTED;                                              \
  }
  if (c->variant == TEMP_GRU_TORCH) { TEMP_CHAIN_FWD(TEMP_GRU_TORCH) }
  TEMP_CHAIN_FWD(TEMP_GRU_TYPE1)
#undef TEMP_CHAIN_FWD
}

int temp_gru_chain_bwd_g4(const TempGruChain* c, const float* saved, int n_up, const float* const* up, float* g4, void* stream) {
  int rc = chain_check(c);
  if (rc) return rc;
  if (n_up < 0 || n_up > TEMP_CHAIN_MAX_UP || (n_up > 0 && !up)) return TEMP_E_BADARG;
  if (c->variant != TEMP_GRU_TORCH) return TEMP_E_UNSUPPORTED;        // (the type-1 cell's dgi is [n, d])
  if (c->n_panels == 0) return TEMP_OK;
  if (!saved || !g4) return TEMP_E_BADARG;
  const ChainArgs a = chain_args(c);
  ChainUps ups = {};
  for (int i = 0; i < n_up; ++i) ups.p[i] = up[i];
  hipStream_t st = (hipStream_t)stream;
  const int tpw = ceil_div(chain_geom(c->d).NTb, 4);
  if (tpw == 1) return launch_chain_bwd<TEMP_GRU_TORCH, 1, 1>(a, ups, saved, g4, nullptr, st);
  if (tpw == 2) return launch_chain_bwd<TEMP_GRU_TORCH, 2, 1>(a, ups, saved, g4, nullptr, st);
  return TEMP_E_UNSUPPORTED;
}

int temp_gru_chain_keys_supported(int d) { return d > 0 && chain_hx(d) ? 1 : 0; }

int temp_gru_chain_bwd_g4_keys(const TempGruChain* c, const float* saved, int n_up, const float* const* up, float* g4, uint32_t* row_keys,
                               uint32_t* col_keys, void* stream) {
  int rc = chain_check(c);
  if (rc) return rc;
  if (n_up < 0 || n_up > TEMP_CHAIN_MAX_UP || (n_up > 0 && !up)) return TEMP_E_BADARG;
  if (c->variant != TEMP_GRU_TORCH || !chain_hx(c->d)) return TEMP_E_UNSUPPORTED;
  if (c->n_panels == 0) return TEMP_OK;
  if (!saved || !g4) return TEMP_E_BADARG;
  const ChainArgs a = chain_args(c);
  ChainUps ups = {};
  for (int i = 0; i < n_up; ++i) ups.p[i] = up[i];
  hipStream_t st = (hipStream_t)stream;
  const int tpw = ceil_div(chain_geom(c->d).NTb, 4);
  if (tpw == 1) return launch_chain_bwd<TEMP_GRU_TORCH, 1, 1>(a, ups, saved, g4, nullptr, st, row_keys, col_keys);
  if (tpw == 2) return launch_chain_bwd<TEMP_GRU_TORCH, 2, 1>(a, ups, saved, g4, nullptr, st, row_keys, col_keys);
  return TEMP_E_UNSUPPORTED;
}

int temp_gru_chain_bwd(const TempGruChain* c, const float* saved, int n_up, const float* const* up, float* dgi, float* dgh, void* stream) {
  int rc = chain_check(c);
  if (rc) return rc;
  if (n_up < 0 || n_up > TEMP_CHAIN_MAX_UP || (n_up > 0 && !up)) return TEMP_E_BADARG;
  if (c->n_panels == 0) return TEMP_OK;
  if (!saved || !dgi || !dgh) return TEMP_E_BADARG;
  const ChainArgs a = chain_args(c);
  ChainUps ups = {};
  for (int i = 0; i < n_up; ++i) ups.p[i] = up[i];
  hipStream_t st = (hipStream_t)stream;
  const int tpw = ceil_div(chain_geom(c->d).NTb, 4);
  if (c->variant == TEMP_GRU_TORCH) {
    if (tpw == 1) return launch_chain_bwd<TEMP_GRU_TORCH, 1>(a, ups, saved, dgi, dgh, st);
    if (tpw == 2) return launch_chain_bwd<TEMP_GRU_TORCH, 2>(a, ups, saved, dgi, dgh, st);
  } else {
    if (tpw == 1) return launch_chain_bwd<TEMP_GRU_TYPE1, 1>(a, ups, saved, dgi, dgh, st);
    if (tpw == 2) return launch_chain_bwd<TEMP_GRU_TYPE1, 2>(a, ups, saved, dgi, dgh, st);
  }
  return TEMP_E_UNSUPPORTED;
}

}  // extern "C"
