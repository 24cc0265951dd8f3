// Window-chain FORWARD, third form (round 6): no matrix / memory roles.  A wave owns ONE unit tile -- 32 hidden units -- and with it
// the three gate-column tiles (r, z, n) of those units: W_hh's gate blocks are packed padded to a whole number of tiles, so the three
// pre-activations of a (track, unit) pair land in the SAME lane.  The wave multiplies (f16 two-way split, split_f16.hpp: 9 MFMAs per
// slab), applies the gates to its own accumulators, keeps its units' fp32 state in registers, writes the saved planes and puts the
// split state back into LDS for everybody's next position.
//
// What that buys over k_gru_chain_fwd_hx (8 matrix + 8 memory waves):
//  * no hand-over of the 32 x 3d products through LDS (78 KB) and no second barrier per position: the state planes are double
//    buffered (position s reads buffer s & 1, writes the other);
//  * 77 KB of LDS and 8 waves of 128 registers per workgroup: TWO workgroups per CU -- one panel's gate / store phase runs beside the
//    other panel's products (inside one workgroup the phases of a position are serial by data dependence);
//  * price: 21 column tiles instead of 19 at d = 200 (the padding of the three blocks), row-scattered 16-byte global accesses (a
//    lane owns one track; L2 merges the four quads of a tile row into full lines).
#pragma once
#include "gru_chain_hx.hpp"

namespace temp {

struct ChainGeomV3 {
  int UT, NT, NS;    // unit tiles (ceil(d / 32)), packed gate-column tiles (3 UT), slabs of 16 k
  int ldp;           // bytes: rows of the state planes (an odd number of 16-byte units)
};
__host__ __device__ inline ChainGeomV3 chain_geom_v3(int D) {
  ChainGeomV3 g;
  g.UT = (D + 31) >> 5; g.NT = 3 * g.UT; g.NS = (D + 15) >> 4;
  g.ldp = g.NS * 32 + 16;
  return g;
}
#define CHV3_WAVES 8
inline size_t chain_lds_fwd_v3(int D, int ms) {
  const ChainGeomV3 g = chain_geom_v3(D);
  return 4 * (size_t)CH_SLOTS * g.ldp + 2 * (size_t)g.NT * 32 * 4 + (2 * CH_SLOTS + 1) * (size_t)ms * 4;
}
inline size_t chain_v3_items(int D) { const ChainGeomV3 g = chain_geom_v3(D); return (size_t)g.NS * g.NT * 128; }
inline size_t chain_v3_pack_floats(int D) { const ChainGeomV3 g = chain_geom_v3(D); return chain_v3_items(D) * 4 + (size_t)g.NT * 32; }

// R.wf / R.kf here: the PADDED forward planes and their keys (packed column T * 32 + c, T = gate * UT + u  <->  gate column gate * d + 32 u + c)
template <int VARIANT>
__global__ void __launch_bounds__(CHV3_WAVES * 64, 4) k_gru_chain_fwd_v3(ChainArgs a, const float* __restrict__ gi, float* __restrict__ H,
                                                                          float* __restrict__ saved) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = a.D;
  const int G = (VARIANT == TEMP_GRU_TORCH) ? 3 * D : D;
  const ChainGeomV3 g = chain_geom_v3(D);
  const int UT = g.UT, NT = g.NT, NS = g.NS, ldp = g.ldp;
  char* hpl = (char*)lds;                                        // [2 buffers][2 planes][32][ldp bytes]  the state, split
  float* ivt = (float*)(hpl + 4 * CH_SLOTS * ldp);               // [NT * 32]  per packed gate column: 1 / (column scale . state scale)
  float* bht = ivt + NT * 32;                                    // [NT * 32]  b_hh in packed column order
  int* tabb = (int*)(bht + NT * 32);                             // [ms][32] the panel's row table
  float* decb = (float*)(tabb + CH_SLOTS * a.max_steps);         // [ms][32] decay factor of every row
  int* flagb = (int*)(decb + CH_SLOTS * a.max_steps);            // [ms] step flags
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, hh = lane >> 5;
  const size_t plane = a.plane;
  const int plane_b = CH_SLOTS * ldp, buf_b = 2 * CH_SLOTS * ldp;
  const bool active = wave < UT;                                 // (wave-uniform) this wave owns unit tile `wave`
  const int u = active ? wave : 0;

  for (int p = blockIdx.x; p < a.n_panels; p += gridDim.x) {
    const int rnn_id = a.panel[4 * p], s0 = a.panel[4 * p + 1], ns = a.panel[4 * p + 2];
    const ChainRnn R = a.rnn[rnn_id];
    for (int i = tid; i < 4 * CH_SLOTS * ldp / 16; i += blockDim.x) reinterpret_cast<hx_u32x4*>(hpl)[i] = hx_u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < NT * 32; i += blockDim.x) {
      ivt[i] = hx_inv_scale(R.kf[i]) * CHX_STATE_INV;
      const int gate = i / (UT * 32), w = i - gate * (UT * 32);
      bht[i] = w < D ? R.b_hh[gate * D + w] : 0.f;
    }
    for (int i = tid; i < ns * CH_SLOTS; i += blockDim.x) {
      const int e = a.rows[(size_t)s0 * CH_SLOTS + i];
      tabb[i] = e;
      decb[i] = e >= 0 ? expf(-a.dt[e & CH_ROW_MASK] * a.lambda) : 0.f;
    }
    if (tid < ns) flagb[tid] = a.sinfo[4 * (size_t)(s0 + tid)];
    __syncthreads();

    // the fp32 state of (track li, units 32 u + 8 q + 4 hh .. + 3), q = 0 .. 3: this lane's, position after position
    float4 hst[4] = {zero4(), zero4(), zero4(), zero4()};
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const hx_u32x4* wp = reinterpret_cast<const hx_u32x4*>(R.wf);
    hx_u32x4 w[2][3];                                            // [plane h, l][gate r, z, n]: ONE slab (16 waves per CU cover the L2 latency)
    auto wload = [&](int sl, int pl) {
#pragma unroll
      for (int j = 0; j < 3; ++j) w[pl][j] = wp[((size_t)(sl * NT + j * UT + u) * 2 + pl) * 64 + lane];
    };
    const int rot = (int)(blockIdx.x >> 3) % NS;                 // per-block start of the slab walk (fixed per block: bit-repeatable)

    for (int s = 0; s < ns; ++s) {
      const int flags = flagb[s];
      const char* hcur = hpl + (s & 1) * buf_b;
      char* hnxt = hpl + ((s + 1) & 1) * buf_b;
      if (active) {
        if (flags & 1) {
          const char* hrow = hcur + (size_t)li * ldp + 16 * hh;  // + plane_b for l, + 32 slab: k = 16 slab + 8 hh .. + 7 of track li
          // (the first slab's planes are asked for HERE, not behind the previous position's last slab: their 24 registers are free
          //  during the gates -- 128 per wave -- and the other fifteen waves of the CU cover the round trip)
          wload(rot, 0); wload(rot, 1);
          for (int j = 0, sl = rot; j < NS; ++j) {
            const int sn = sl + 1 < NS ? sl + 1 : 0;
            const hx_f16x8 ah = hx_frag(*reinterpret_cast<const hx_u32x4*>(hrow + 32 * sl));
            const hx_f16x8 al = hx_frag(*reinterpret_cast<const hx_u32x4*>(hrow + plane_b + 32 * sl));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx_frag(w[1][t]), ah, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < NS) wload(sn, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx_frag(w[0][t]), al, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx_frag(w[0][t]), ah, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < NS) wload(sn, 0);
            sl = sn;
          }
        }
        // ---- gates of this lane's (track, unit) pairs: quad q = units 32 u + 8 q + 4 hh .. + 3
        const int e = tabb[s * CH_SLOTS + li];
        const bool live = e >= 0;
        const bool hp = live && (e & CH_HAS_PREV);
        const size_t row = live ? (size_t)(e & CH_ROW_MASK) : 0;
        const float dec = hp ? decb[s * CH_SLOTS + li] : 0.f;    // (a track without a previous state: its products are multiplied away)
        const float* gsrc = gi + (size_t)(a.gi_index ? a.gi_index[row] : (int)row) * G;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_barrier(0);                     // (one quad's loads and temporaries at a time: 128 registers)
          const int cu = 32 * u + 8 * q + 4 * hh;                // first unit of the quad
          const bool ok = live && cu < D;
          const int cc = ok ? cu : 0;
          float4 g0 = zero4(), g1 = zero4(), g2 = zero4();
          if (VARIANT == TEMP_GRU_TORCH) { g0 = ld4(gsrc + cc); g1 = ld4(gsrc + D + cc); g2 = ld4(gsrc + 2 * D + cc); }
          else g2 = ld4(gsrc + cc);
          const int pc = 32 * u + 8 * q + 4 * hh;                // packed column inside a gate block
          const float4 ir = ld4(ivt + pc), iz = ld4(ivt + UT * 32 + pc), in_ = ld4(ivt + 2 * UT * 32 + pc);
          const float4 br = ld4(bht + pc), bz = ld4(bht + UT * 32 + pc), bn = ld4(bht + 2 * UT * 32 + pc);
          const float arv[4] = {acc[0][4 * q] * (ir.x * dec), acc[0][4 * q + 1] * (ir.y * dec), acc[0][4 * q + 2] * (ir.z * dec), acc[0][4 * q + 3] * (ir.w * dec)};
          const float azv[4] = {acc[1][4 * q] * (iz.x * dec), acc[1][4 * q + 1] * (iz.y * dec), acc[1][4 * q + 2] * (iz.z * dec), acc[1][4 * q + 3] * (iz.w * dec)};
          const float anv[4] = {acc[2][4 * q] * (in_.x * dec), acc[2][4 * q + 1] * (in_.y * dec), acc[2][4 * q + 2] * (in_.z * dec), acc[2][4 * q + 3] * (in_.w * dec)};
#pragma unroll
          for (int t = 0; t < 3; ++t) { acc[t][4 * q] = 0.f; acc[t][4 * q + 1] = 0.f; acc[t][4 * q + 2] = 0.f; acc[t][4 * q + 3] = 0.f; }
          const float4 hd4 = scale4(hst[q], dec);                // decayed previous state (models/RRGCN.py:83)
          const float hdv[4] = {hd4.x, hd4.y, hd4.z, hd4.w};
          const float g0v[4] = {g0.x, g0.y, g0.z, g0.w}, g1v[4] = {g1.x, g1.y, g1.z, g1.w}, g2v[4] = {g2.x, g2.y, g2.z, g2.w};
          const float brv[4] = {br.x, br.y, br.z, br.w}, bzv[4] = {bz.x, bz.y, bz.z, bz.w}, bnv[4] = {bn.x, bn.y, bn.z, bn.w};
          float o_h[4], o_r[4], o_z[4], o_n[4], o_hn[4];
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
          if (ok) {
            const float4 h4 = make_float4(o_h[0], o_h[1], o_h[2], o_h[3]);
            hst[q] = h4;
            hx_u32x2 SH, SL;
            hx_split4(h4, CHX_STATE_SCALE, SH, SL);
            char* hdst = hnxt + (size_t)li * ldp + 2 * cu;
            *reinterpret_cast<hx_u32x2*>(hdst) = SH;
            *reinterpret_cast<hx_u32x2*>(hdst + plane_b) = SL;
            const size_t o = row * D + cu;
            if (flags & 2) st4(H + o, h4);
            st4(saved + o, make_float4(o_r[0], o_r[1], o_r[2], o_r[3]));
            st4(saved + plane + o, make_float4(o_z[0], o_z[1], o_z[2], o_z[3]));
            st4(saved + 2 * plane + o, make_float4(o_n[0], o_n[1], o_n[2], o_n[3]));
            st4(saved + 3 * plane + o, make_float4(o_hn[0], o_hn[1], o_hn[2], o_hn[3]));
            st4(saved + 4 * plane + o, hd4);
          }
        }
      }
      __syncthreads();        // the split state of position s is in buffer (s + 1) & 1; buffer s & 1 is free to be overwritten at s + 1
    }
    __syncthreads();          // LDS is re-initialised for the next panel
  }
}

}  // namespace temp
