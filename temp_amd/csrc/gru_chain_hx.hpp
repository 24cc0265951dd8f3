// The window-chain kernels (gru_chain.hip) with their W_hh products on the f16 matrix pipe: three MFMA products of the scaled
// two-way operand split (split_f16.hpp) where the bf16 variant issues six, two weight planes streamed from L2 instead of three.
//
// What changes against k_gru_chain_fwd / _bwd<.., BX = 1>:
//  * The activations-side operand is split ONCE, by the memory-role wave that produces it, and lives in LDS as two f16 planes in
//    row-major order (track, k): the matrix waves read their MFMA fragments (8 consecutive k of track li) with one ds_read_b128
//    per plane and issue no VALU work at all (the bf16 variant split the fp32 panel in every one of the four matrix waves).
//      forward : the state h.  |h| <= 1 (a convex combination of tanh outputs, starting from 0), so the scale is the constant 2^14;
//                the decay of the previous state is a per-track factor and moves BEHIND the product: (dec h) W = dec (h W).
//                The fp32 state itself (the gates blend it) stays in LDS beside the planes, updated in place by the lane that owns it.
//      backward: the gate gradients [dr | dz | dn_h] of a track, scaled by the power of two that puts the row's largest magnitude into
//                [2^14, 2^15) (a wave holds a whole row: one DPP reduction); the matrix waves unscale d_prev.
//  * W_hh arrives as two f16 planes with per-column scales (hx_pack.hpp); the gates / d_prev epilogues unscale per column.
//  * The backward also hands out what the consumers of g4 need for THEIR split (temp_gru_grads_g4): per row the key of
//    max |[dr dz dn_i]| (d_x = g4[:, :3d] . W_ih) and per GRU and column of g4 the key of the column maximum (the weight gradients sum
//    over the rows: their scale is per column) -- integer maxima, order-independent, so the result is bit-repeatable.
#pragma once
#include "hx_pack.hpp"

namespace temp {

#define CHX_STATE_SCALE 16384.f
#define CHX_STATE_INV (1.f / 16384.f)

struct ChainGeomHx {
  int NT, NS;        // forward: tiles of 32 gate columns (3d), slabs of 16 k (d)
  int NTb, NSb;      // backward: tiles of 32 state columns (d), slabs of 16 k (3d) rounded up to a multiple of 8 (zero slabs)
  int lda, ldz;      // floats: forward product rows; backward gz / d_prev rows
  int ldp, ldpa;     // bytes: rows of the forward state planes / the backward gate-gradient planes (an odd number of 16-byte units)
  int ldh;           // floats: forward fp32 state rows
};
__host__ __device__ inline ChainGeomHx chain_geom_hx(int D) {
  ChainGeomHx g;
  g.NT = (3 * D + 31) >> 5; g.NS = (D + 15) >> 4;
  g.NTb = (D + 31) >> 5; g.NSb = (((3 * D + 15) >> 4) + 7) & ~7;      // (a multiple of the W ring of either wave configuration)
  g.lda = g.NT * 32 + 4; g.ldz = g.NTb * 32 + 4;
  g.ldp = g.NS * 32 + 16; g.ldpa = g.NSb * 32 + 16;
  g.ldh = g.NS * 16 + 4;
  return g;
}
inline size_t chain_lds_fwd_hx(int D, int ms) {
  const ChainGeomHx g = chain_geom_hx(D);
  return (size_t)CH_SLOTS * g.lda * 4 + 2 * (size_t)CH_SLOTS * g.ldp + (size_t)CH_SLOTS * g.ldh * 4 + (size_t)g.NT * 32 * 4 + (2 * CH_SLOTS + 1) * (size_t)ms * 4;
}
inline size_t chain_lds_bwd_hx(int D, int ms) {
  const ChainGeomHx g = chain_geom_hx(D);
  return 2 * (size_t)CH_SLOTS * g.ldpa + 2 * (size_t)CH_SLOTS * g.ldz * 4 + (2 * CH_SLOTS + 3) * (size_t)ms * 4 + CH_SLOTS * 4 + 4 * (size_t)D * 4;
}
// packed W_hh of one GRU (16-byte items): [forward planes | backward planes | forward keys (NT * 32) | backward keys (NTb * 32)]
inline size_t chain_hx_fwd_items(int D) { const ChainGeomHx g = chain_geom_hx(D); return (size_t)g.NS * g.NT * 128; }
inline size_t chain_hx_bwd_items(int D) { const ChainGeomHx g = chain_geom_hx(D); return (size_t)g.NSb * g.NTb * 128; }
inline size_t chain_hx_pack_floats(int D) {
  const ChainGeomHx g = chain_geom_hx(D);
  return (chain_hx_fwd_items(D) + chain_hx_bwd_items(D)) * 4 + (size_t)(g.NT + g.NTb) * 32;
}

// ---- forward --------------------------------------------------------------------------------------------------------
// NMW matrix waves (4 or 8: one or two per SIMD) with TPW = ceil(NT / NMW) tiles each, MW memory waves
// TWOSETS (NMW = 8 only): two register sets of W_hh planes instead of one, paid for by reading the state fragments just in time
template <int VARIANT, int TPW, int MW, int NMW = 4, int TWOSETS = 0>
__global__ void __launch_bounds__(64 * (NMW + MW)) k_gru_chain_fwd_hx(ChainArgs a, const float* __restrict__ gi, float* __restrict__ H,
                                                                       float* __restrict__ saved) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PASSES = CH_SLOTS / MW;
  const int D = a.D, D4 = D >> 2;
  const int G = (VARIANT == TEMP_GRU_TORCH) ? 3 * D : D;
  const ChainGeomHx g = chain_geom_hx(D);
  const int NT = g.NT, NS = g.NS, lda = g.lda, ldp = g.ldp, ldh = g.ldh;
  float* accb = lds;                                             // [32][lda]  raw (scaled) products of the current position
  char* hpl = (char*)(accb + CH_SLOTS * lda);                    // [2 planes][32][ldp bytes]  the state, split
  float* hb = (float*)(hpl + 2 * CH_SLOTS * ldp);                // [32][ldh]  the state in fp32 (read and rewritten by its owner lane only)
  float* ivt = hb + CH_SLOTS * ldh;                              // [NT * 32]  per gate column: 1 / (column scale . state scale)
  int* tabb = (int*)(ivt + NT * 32);                             // [ms][32] the panel's row table
  float* decb = (float*)(tabb + CH_SLOTS * a.max_steps);         // [ms][32] decay factor of every row
  int* flagb = (int*)(decb + CH_SLOTS * a.max_steps);            // [ms] step flags
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;    // (wave-uniform: scalar tile indices, scalar branches)
  const size_t plane = a.plane;

  for (int p = blockIdx.x; p < a.n_panels; p += gridDim.x) {
    const int rnn_id = a.panel[4 * p], s0 = a.panel[4 * p + 1], ns = a.panel[4 * p + 2];
    const ChainRnn R = a.rnn[rnn_id];
    for (int i = tid; i < 2 * CH_SLOTS * ldp / 16; i += blockDim.x) reinterpret_cast<hx_u32x4*>(hpl)[i] = hx_u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < CH_SLOTS * ldh; i += blockDim.x) hb[i] = 0.f;
    for (int i = tid; i < NT * 32; i += blockDim.x) ivt[i] = hx_inv_scale(R.kf[i]) * CHX_STATE_INV;
    for (int i = tid; i < ns * CH_SLOTS; i += blockDim.x) {
      const int e = a.rows[(size_t)s0 * CH_SLOTS + i];
      tabb[i] = e;
      decb[i] = e >= 0 ? expf(-a.dt[e & CH_ROW_MASK] * a.lambda) : 0.f;
    }
    if (tid < ns) flagb[tid] = a.sinfo[4 * (size_t)(s0 + tid)];
    __syncthreads();

    if (wave < NMW) {
      // ------------------------------------------------------------------ matrix role: pure MFMA + fragment reads
      const int li = lane & 31, hh = lane >> 5;
      bool tval[TPW];
      int tidx[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) { tidx[j] = wave + NMW * j; tval[j] = tidx[j] < NT; if (!tval[j]) tidx[j] = NT - 1; }
      f32x16 acc[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      // W_hh planes of TWO slabs in registers (sets 0 / 1 alternate along the slab walk; a set is refilled with the slab two ahead as
      // soon as its plane has had its last product: L after round 0, H after round 2).
      const hx_u32x4* wp = reinterpret_cast<const hx_u32x4*>(R.wf);
      constexpr bool ONESET = NMW > 4 && !TWOSETS;
      constexpr bool PREF = !(NMW > 4 && TWOSETS);               // read the next slab's state fragments one slab ahead
      hx_u32x4 w[ONESET ? 1 : 2][2][TPW] = {};                   // [set][plane h, l][tile]
      auto wload = [&](hx_u32x4 (&wr)[TPW], int sl, int pl) {
#pragma unroll
        for (int j = 0; j < TPW; ++j)
          if (tval[j]) wr[j] = wp[((size_t)(sl * NT + tidx[j]) * 2 + pl) * 64 + lane];
      };
      // every block starts its walk at its own slab (the CUs of an XCD do not ask L2 for the same lines at the same time; fixed
      // per block, so results stay bit-repeatable)
      const int rot = (a.dbg & 64) ? 0 : (int)(blockIdx.x >> 3) % NS;
      wload(w[0][0], rot, 0); wload(w[0][1], rot, 1);
      if constexpr (!ONESET) { wload(w[1][0], rot + 1 < NS ? rot + 1 : 0, 0); wload(w[1][1], rot + 1 < NS ? rot + 1 : 0, 1); }
      const char* hrow = hpl + (size_t)li * ldp + 16 * hh;       // + plane * 32 ldp + 32 slab: k = 16 slab + 8 hh .. + 7 of track li
      const int pl1 = CH_SLOTS * ldp;
      for (int s = 0; s < ns; ++s) {
        const int flags = flagb[s];
        if ((flags & 1) && !(a.dbg & 1)) {                      // (dbg bit 0: development ablation, no products)
          hx_u32x4 FH = *reinterpret_cast<const hx_u32x4*>(hrow + 32 * rot), FL = *reinterpret_cast<const hx_u32x4*>(hrow + pl1 + 32 * rot), NH, NL;
          int sl_cur = rot;
          // one slab out of register set SET (a compile-time index: the sets are registers, never addressed); sl1: the next slab
          // (its fragments are read now), sl2: the slab the set is refilled with
          auto slab = [&](auto set_c, int sl1, int sl2) {
            constexpr int SET = decltype(set_c)::value;
            if constexpr (PREF) {
              NH = *reinterpret_cast<const hx_u32x4*>(hrow + 32 * sl1);
              NL = *reinterpret_cast<const hx_u32x4*>(hrow + pl1 + 32 * sl1);
            } else {
              FH = *reinterpret_cast<const hx_u32x4*>(hrow + 32 * sl_cur);
              FL = *reinterpret_cast<const hx_u32x4*>(hrow + pl1 + 32 * sl_cur);
              sl_cur = sl1;
            }
            const hx_f16x8 ah = hx_frag(FH), al = hx_frag(FL);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx_frag(w[SET][1][j]), ah, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wload(w[SET][1], sl2, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx_frag(w[SET][0][j]), al, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx_frag(w[SET][0][j]), ah, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wload(w[SET][0], sl2, 0);
            if constexpr (PREF) { FH = NH; FL = NL; }
          };
          // Walk position j = 0 .. NS - 1 visits slab (rot + j) mod NS out of set j & 1.  A set is refilled with the slab of walk
          // position j + 2 -- or, behind the walk's last two slabs, with walk position j & 1 of the NEXT position (set 0 always holds
          // the even walk positions, also for an odd slab count: the loads behind the last slabs have the whole gate phase to land).
          auto at = [&](int j) { const int v = rot + j; return v < NS ? v : v - NS; };
          if constexpr (ONESET) {
            // two matrix waves per SIMD cover each other's L2 latency: ONE register set, refilled in place with the next slab
            for (int j = 0; j < NS; ++j) slab(std::integral_constant<int, 0>(), at(j + 1 < NS ? j + 1 : 0), at(j + 1 < NS ? j + 1 : 0));
          } else {
          int j = 0;
          for (; j + 1 < NS; j += 2) {
            slab(std::integral_constant<int, 0>(), at(j + 1), at(j + 2 < NS ? j + 2 : 0));
            slab(std::integral_constant<int, 1>(), at(j + 2 < NS ? j + 2 : 0), at(j + 3 < NS ? j + 3 : 1));
          }
          if (j < NS) slab(std::integral_constant<int, 0>(), at(0), at(0));
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
        __syncthreads();      // B: the split state of position s is in LDS
      }
    } else {
      // ------------------------------------------------------------------ memory role
      const int mw = wave - NMW, c4 = lane, col = 4 * c4;
      const bool cact = c4 < D4;
      const int colc = cact ? col : 0;
      const float4 bhr = ld4(R.b_hh + colc), bhz = ld4(R.b_hh + D + colc), bhn = ld4(R.b_hh + 2 * D + colc);
      int erow[PASSES];
      float4 g0[PASSES], g1[PASSES], g2[PASSES];
      auto prefetch = [&](int s) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int e = tabb[s * CH_SLOTS + ps * MW + mw];
          erow[ps] = e;
          const bool ok = e >= 0 && cact && !(a.dbg & 4);       // (dbg bit 2: development ablation, no input-gate loads)
          const int er = e & CH_ROW_MASK;
          const float* src = gi + (ok ? (size_t)(a.gi_index ? a.gi_index[er] : er) * G + col : 0);
          if (VARIANT == TEMP_GRU_TORCH) { g0[ps] = ld4(src); g1[ps] = ld4(src + (ok ? D : 0)); g2[ps] = ld4(src + (ok ? 2 * D : 0)); }
          else { g0[ps] = zero4(); g1[ps] = zero4(); g2[ps] = ld4(src); }
        }
      };
      prefetch(0);
      for (int s = 0; s < ns; ++s) {
        const int flags = flagb[s];
        __syncthreads();      // A
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int slot = ps * MW + mw;
          const int e = erow[ps];
          if (e < 0 || !cact) continue;
          const size_t row = (size_t)(e & CH_ROW_MASK);
          const bool hp = (e & CH_HAS_PREV) != 0;
          float4 ar = zero4(), az = zero4(), an = zero4(), hd = zero4();
          if (hp) {                                   // (a track without a previous state: whatever its planes held is not read)
            const float* ab = accb + (size_t)slot * lda + col;
            const float dec = decb[s * CH_SLOTS + slot];
            const float4 pr = ld4(ab), pz = ld4(ab + D), pn = ld4(ab + 2 * D);
            const float4 ivr = ld4(ivt + col), ivz = ld4(ivt + D + col), ivn = ld4(ivt + 2 * D + col);
            ar = make_float4(pr.x * (ivr.x * dec), pr.y * (ivr.y * dec), pr.z * (ivr.z * dec), pr.w * (ivr.w * dec));
            az = make_float4(pz.x * (ivz.x * dec), pz.y * (ivz.y * dec), pz.z * (ivz.z * dec), pz.w * (ivz.w * dec));
            an = make_float4(pn.x * (ivn.x * dec), pn.y * (ivn.y * dec), pn.z * (ivn.z * dec), pn.w * (ivn.w * dec));
            hd = scale4(ld4(hb + (size_t)slot * ldh + col), dec);   // decayed previous state (models/RRGCN.py:83)
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
          st4(hb + (size_t)slot * ldh + col, h4);
          hx_u32x2 SH, SL;
          hx_split4(h4, CHX_STATE_SCALE, SH, SL);
          char* hdst = hpl + (size_t)slot * ldp + 2 * col;
          *reinterpret_cast<hx_u32x2*>(hdst) = SH;
          *reinterpret_cast<hx_u32x2*>(hdst + CH_SLOTS * ldp) = SL;
          const size_t o = row * D + col;
          if (a.dbg & 2) continue;                              // (development ablation: no global stores)
          if (flags & 2) st4(H + o, h4);
          st4(saved + o, make_float4(o_r[0], o_r[1], o_r[2], o_r[3]));
          st4(saved + plane + o, make_float4(o_z[0], o_z[1], o_z[2], o_z[3]));
          st4(saved + 2 * plane + o, make_float4(o_n[0], o_n[1], o_n[2], o_n[3]));
          st4(saved + 3 * plane + o, make_float4(o_hn[0], o_hn[1], o_hn[2], o_hn[3]));
          st4(saved + 4 * plane + o, hd);
        }
        __syncthreads();      // B
        if (s + 1 < ns) prefetch(s + 1);       // in flight while the matrix waves run position s + 1
      }
    }
    __syncthreads();          // LDS is re-initialised for the next panel
  }
}

// ---- backward -------------------------------------------------------------------------------------------------------
// row_keys (nullable): [N_total] key of max |[dr dz dn_i]| of every row; col_keys (nullable): [n_rnn + n_panels][4d]: the kernel writes row
// n_rnn + p (panel p's column maxima), k_keys_reduce (hx_pack.hpp) reduces them into rows 0 .. n_rnn - 1
// NMW matrix waves (4 or 8: one or two per SIMD) with TPWB = ceil(NTb / NMW) tiles each; RING = slabs of W_hh held in registers
template <int VARIANT, int TPWB, int MW, int G4, int NMW = 4, int RING = 4>
__global__ void __launch_bounds__(64 * (NMW + MW)) k_gru_chain_bwd_hx(ChainArgs a, ChainUps ups, const float* __restrict__ saved,
                                                                     float* __restrict__ dgi, float* __restrict__ dgh,
                                                                     unsigned* __restrict__ row_keys, unsigned* __restrict__ col_keys) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PASSES = CH_SLOTS / MW;
  const int D = a.D, D4 = D >> 2;
  const ChainGeomHx g = chain_geom_hx(D);
  const int NTb = g.NTb, NSb = g.NSb, ldpa = g.ldpa, ldz = g.ldz;
  char* apl = (char*)lds;                                        // [2 planes][32][ldpa bytes]  [dr | dz | dn_h] of the position, scaled per row, split
  float* gzb = (float*)(apl + 2 * CH_SLOTS * ldpa);              // [32][ldz]  dh * z
  float* dpb = gzb + CH_SLOTS * ldz;                             // [32][ldz]  d_prev of the position just processed
  int* tabb = (int*)(dpb + CH_SLOTS * ldz);                      // [ms][32] the panel's row table
  float* decb = (float*)(tabb + CH_SLOTS * a.max_steps);         // [ms][32]
  int* flagb = (int*)(decb + CH_SLOTS * a.max_steps);            // [ms]
  int* upb = flagb + a.max_steps;                                // [ms][2] upstream block and its first row of every step
  float* rinv = (float*)(upb + 2 * a.max_steps);                 // [32] 1 / row scale of the position's planes
  unsigned* ckey = (unsigned*)(rinv + CH_SLOTS);                 // [4d] column maxima of this panel's g4 rows
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;    // (wave-uniform: scalar tile indices, scalar branches)
  const size_t plane = a.plane;

  for (int p = blockIdx.x; p < a.n_panels; p += gridDim.x) {
    const int rnn_id = a.panel[4 * p], s0 = a.panel[4 * p + 1], ns = a.panel[4 * p + 2];
    const ChainRnn R = a.rnn[rnn_id];
    for (int i = tid; i < 2 * CH_SLOTS * ldpa / 16; i += blockDim.x) reinterpret_cast<hx_u32x4*>(apl)[i] = hx_u32x4{0u, 0u, 0u, 0u};   // k padding
    for (int i = tid; i < 4 * D; i += blockDim.x) ckey[i] = 0u;
    if (tid < ns) { upb[2 * tid] = a.sinfo[4 * (size_t)(s0 + tid) + 1]; upb[2 * tid + 1] = a.sinfo[4 * (size_t)(s0 + tid) + 2]; }
    for (int i = tid; i < ns * CH_SLOTS; i += blockDim.x) {
      const int e = a.rows[(size_t)s0 * CH_SLOTS + i];
      tabb[i] = e;
      decb[i] = e >= 0 ? expf(-a.dt[e & CH_ROW_MASK] * a.lambda) : 0.f;
    }
    if (tid < ns) flagb[tid] = a.sinfo[4 * (size_t)(s0 + tid)];
    __syncthreads();

    if (wave < NMW) {
      // ------------------------------------------------------------------ matrix role: d_prev = (dgh . W_hh + dh*z) * decay
      const int li = lane & 31, hh = lane >> 5;
      bool tval[TPWB];
      int tidx[TPWB];
#pragma unroll
      for (int j = 0; j < TPWB; ++j) { tidx[j] = wave + NMW * j; tval[j] = tidx[j] < NTb; if (!tval[j]) tidx[j] = NTb - 1; }
      f32x16 acc[TPWB];
#pragma unroll
      for (int j = 0; j < TPWB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      // W_hh planes of FOUR slabs in registers (a slab is only 3 TPWB products): set q holds slab 4 i + q and is refilled with slab
      // 4 i + q + 4 as its planes fall free -- three slabs (18 products at TPWB = 2) of L2 latency cover.  NSb is a multiple of 4.
      const hx_u32x4* wp = reinterpret_cast<const hx_u32x4*>(R.wb);
      hx_u32x4 wh[RING][TPWB] = {}, wl[RING][TPWB] = {};
      auto wload = [&](hx_u32x4 (&w)[TPWB], int sl, int pl) {
#pragma unroll
        for (int j = 0; j < TPWB; ++j)
          if (tval[j]) w[j] = wp[((size_t)(sl * NTb + tidx[j]) * 2 + pl) * 64 + lane];
      };
      const int rot = (a.dbg & 64) ? 0 : RING * ((int)(blockIdx.x >> 3) % (NSb / RING));     // per-block start of the slab walk (NSb % RING == 0)
#pragma unroll
      for (int q = 0; q < RING; ++q) { wload(wh[q], rot + q, 0); wload(wl[q], rot + q, 1); }
      const char* arow = apl + (size_t)li * ldpa + 16 * hh;      // + plane * 32 ldpa + 32 slab
      const int pl1 = CH_SLOTS * ldpa;
      for (int s = ns - 1; s >= 0; --s) {
        const int flags = flagb[s];
        __syncthreads();      // A: the split gate gradients / dh*z / row scales of position s are in LDS
        if ((flags & 1) && !(a.dbg & 1)) {                      // (dbg bit 0: development ablation, no products)
          hx_u32x4 FH = *reinterpret_cast<const hx_u32x4*>(arow + 32 * rot), FL = *reinterpret_cast<const hx_u32x4*>(arow + pl1 + 32 * rot), NH, NL;
          for (int j = 0, sl = rot; j < NSb; j += RING) {
            const int base2 = sl + RING < NSb ? sl + RING : 0;          // (after the walk's last group: the first group of the NEXT position)
#pragma unroll
            for (int q = 0; q < RING; ++q) {
              const int sn = (q < RING - 1) ? sl + q + 1 : base2;
              NH = *reinterpret_cast<const hx_u32x4*>(arow + 32 * sn);
              NL = *reinterpret_cast<const hx_u32x4*>(arow + pl1 + 32 * sn);
              const hx_f16x8 ah = hx_frag(FH), al = hx_frag(FL);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int t = 0; t < TPWB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx_frag(wl[q][t]), ah, acc[t], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              wload(wl[q], base2 + q, 1);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int t = 0; t < TPWB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx_frag(wh[q][t]), al, acc[t], 0, 0, 0);
#pragma unroll
              for (int t = 0; t < TPWB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx_frag(wh[q][t]), ah, acc[t], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              wload(wh[q], base2 + q, 0);
              FH = NH; FL = NL;
            }
            sl = base2;
          }
          const float dec = decb[s * CH_SLOTS + li], ri = rinv[li];
#pragma unroll
          for (int j = 0; j < TPWB; ++j) {
            if (!tval[j]) continue;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const int c = tidx[j] * 32 + 8 * qq + 4 * hh;
              const float4 gz = ld4(gzb + (size_t)li * ldz + c);
              const float i0 = hx_inv_scale(R.kb[c]) * ri, i1 = hx_inv_scale(R.kb[c + 1]) * ri, i2 = hx_inv_scale(R.kb[c + 2]) * ri, i3 = hx_inv_scale(R.kb[c + 3]) * ri;
              st4(dpb + (size_t)li * ldz + c, make_float4((acc[j][4 * qq] * i0 + gz.x) * dec, (acc[j][4 * qq + 1] * i1 + gz.y) * dec,
                                                           (acc[j][4 * qq + 2] * i2 + gz.z) * dec, (acc[j][4 * qq + 3] * i3 + gz.w) * dec));
              acc[j][4 * qq] = 0.f; acc[j][4 * qq + 1] = 0.f; acc[j][4 * qq + 2] = 0.f; acc[j][4 * qq + 3] = 0.f;
            }
          }
        }
        __syncthreads();      // B: d_prev of position s is in LDS
      }
    } else {
      // ------------------------------------------------------------------ memory role: gate gradients
      const int mw = wave - NMW, c4 = lane, col = 4 * c4;
      const bool cact = c4 < D4;
      const int colc = cact ? col : 0;
      int erow[PASSES];
      bool nxt[PASSES];
      float4 sr[PASSES], sz[PASSES], sn[PASSES], shn[PASSES], shd[PASSES];
      unsigned ck[4][4] = {};                                    // running maxima of this lane's four columns of dr, dz, dn_i, dn_h
      auto prefetch = [&](int s) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int slot = ps * MW + mw;
          const int e = tabb[s * CH_SLOTS + slot];
          erow[ps] = e;
          int en = -1;
          if (s + 1 < ns) en = tabb[(s + 1) * CH_SLOTS + slot];
          nxt[ps] = en >= 0 && (en & CH_HAS_PREV);
          const bool ok = e >= 0 && cact && !(a.dbg & 4);       // (dbg bit 2: development ablation, every lane reads row 0)
          const size_t row = ok ? (size_t)(e & CH_ROW_MASK) : 0;
          const float* src = saved + row * D + (ok ? col : 0);
          sr[ps] = ld4(src); sz[ps] = ld4(src + plane); sn[ps] = ld4(src + 2 * plane); shn[ps] = ld4(src + 3 * plane);
          shd[ps] = ld4(src + 4 * plane);
        }
      };
      prefetch(ns - 1);
      for (int s = ns - 1; s >= 0; --s) {
        const int up_sel = upb[2 * s], up_row0 = upb[2 * s + 1];       // (staged in LDS: two dependent global loads per step otherwise)
        const float* upp = up_sel >= 0 ? ups.p[up_sel] : nullptr;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int slot = ps * MW + mw;
          const int e = erow[ps];
          if (e < 0) continue;                 // (wave-uniform) idle track: whatever its LDS rows hold only reaches its own, unread, d_prev row
          const size_t row = (size_t)(e & CH_ROW_MASK);
          // every lane runs the arithmetic (the row maxima are wave reductions); lanes past the width compute on zeros
          float4 gd = (upp && cact) ? ld4(upp + (row - (size_t)up_row0) * D + col) : zero4();
          if (nxt[ps] && cact) gd = add4(gd, ld4(dpb + (size_t)slot * ldz + col));
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
          const unsigned kr = cact ? hx_abs_bits4(dr_pre) : 0u, kz = cact ? hx_abs_bits4(dz_pre) : 0u;
          const unsigned kn = cact ? hx_abs_bits4(dn_pre) : 0u, kh = cact ? hx_abs_bits4(dhn) : 0u;
          if (cact && col_keys) {
            if constexpr (NMW > 4) {                             // (128 registers per wave: the column maxima go to LDS at once)
              auto upl = [&](int b4, const float4 v) {
                unsigned* k = ckey + b4 * D + col;
                atomicMax(k, hx_abs_bits(v.x)); atomicMax(k + 1, hx_abs_bits(v.y)); atomicMax(k + 2, hx_abs_bits(v.z)); atomicMax(k + 3, hx_abs_bits(v.w));
              };
              upl(0, dr_pre); upl(1, dz_pre); upl(2, dn_pre); upl(3, dhn);
            } else {
              auto upd = [](unsigned (&m)[4], const float4 v) { m[0] = max(m[0], hx_abs_bits(v.x)); m[1] = max(m[1], hx_abs_bits(v.y)); m[2] = max(m[2], hx_abs_bits(v.z)); m[3] = max(m[3], hx_abs_bits(v.w)); };
              upd(ck[0], dr_pre); upd(ck[1], dz_pre); upd(ck[2], dn_pre); upd(ck[3], dhn);
            }
          }
          const unsigned krz = max(kr, kz);
          const unsigned key_h = hx_wave_max(max(krz, kh));      // [dr dz dn_h]: this row of the recurrent product
          const float sc = hx_scale(key_h);
          if (cact) {
            hx_u32x2 SH, SL;
            char* ad = apl + (size_t)slot * ldpa + 2 * col;
            hx_split4(dr_pre, sc, SH, SL); *reinterpret_cast<hx_u32x2*>(ad) = SH; *reinterpret_cast<hx_u32x2*>(ad + CH_SLOTS * ldpa) = SL;
            hx_split4(dz_pre, sc, SH, SL); *reinterpret_cast<hx_u32x2*>(ad + 2 * D) = SH; *reinterpret_cast<hx_u32x2*>(ad + 2 * D + CH_SLOTS * ldpa) = SL;
            hx_split4(dhn, sc, SH, SL); *reinterpret_cast<hx_u32x2*>(ad + 4 * D) = SH; *reinterpret_cast<hx_u32x2*>(ad + 4 * D + CH_SLOTS * ldpa) = SL;
            st4(gzb + (size_t)slot * ldz + col, gz);
          }
          if (lane == 0) rinv[slot] = hx_inv_scale(key_h);
          if (row_keys) {
            const unsigned key_x = hx_wave_max(max(krz, kn));    // [dr dz dn_i]: this row of d_x = g4[:, :3d] . W_ih
            if (lane == 0) row_keys[row] = key_x;
          }
          if (cact && !(a.dbg & 2)) {                           // (dbg bit 1: development ablation, no global stores)
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
        }
        __syncthreads();      // A
        if (s > 0) prefetch(s - 1);            // issued behind the barrier (the matrix waves start at once), in flight while they
        __syncthreads();      // B             // run position s
      }
      if (col_keys && cact && NMW <= 4) {      // this lane's column maxima -> the panel's (LDS integer maxima over the eight waves)
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4)
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicMax(&ckey[b4 * D + colc + e], ck[b4][e]);
      }
    }
    __syncthreads();
    if (col_keys) {
      // this panel's column maxima -> its own row of the partials (k_keys_reduce takes the maxima over a GRU's panels: same-address
      // atomics from 250 workgroups on eight L2s cost ~20 us)
      for (int i = tid; i < 4 * D; i += blockDim.x) col_keys[((size_t)a.n_rnn_keys + p) * 4 * D + i] = ckey[i];
      __syncthreads();
    }
  }
}

}  // namespace temp
