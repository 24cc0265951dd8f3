// Pipelined window-chain kernels (round 4): the same contract, tables and arithmetic as k_gru_chain_fwd / _bwd<.., BX = 1>
// (gru_chain.hip), re-cut so that the two roles of a workgroup overlap INSIDE a position instead of taking turns.
//
// What bounded the round-3 kernels: a position of a panel was product (one matrix wave per SIMD, 12.5 k cycles
// of MFMA issue + the W_hh plane stream) -> workgroup barrier -> gates (four memory waves, ~10 k cycles, matrix waves idle)
// -> barrier: ~40 k cycles against an HBM floor of ~21 k (32 tracks x 9 rows x 800 B per position per CU at 6 TB/s / 256).
// The recurrence only forces  product(s+1) [all gate columns] <- state(s) [all k]  and  gates(s+1) <- product(s+1); but the
// product is a sum over k, and k IS the hidden column the gate phase produces.  So both kernels walk a position in CHUNKS of
// 32 hidden columns:
//   forward   gate waves finish chunk c of position s (h' columns 32c..32c+31 of all 32 tracks) and publish it; the matrix
//             waves run the two k-slabs of chunk c of position s+1's product while the gate waves work on chunk c+1.
//   backward  gate waves finish the gate gradients of chunk c (dgh columns {r,z,n} x 32c..) and publish them; the matrix
//             waves run the six k'-slabs of that chunk of d_prev = dgh . W_hh (k' = chunk-major order of the 3d gate columns,
//             the packed W_hh follows it) while the gate waves work on the next chunk.
// Only the first chunk of the gate phase and the hand-over of the raw products stay exposed per position.
// Hand-over is an LDS counter per direction (ds_add / ds_read polls with s_sleep), not a workgroup barrier: a barrier would
// re-join the two roles at every chunk.  Spins are bounded (a logic error shows up as a wrong result + temp_gru_chain_timeouts()
// > 0, never as a hung GPU).
// The state / dgh operand lives in LDS as the three bf16 planes of the exact split (gemm_bx.hpp) in MFMA fragment order, WRITTEN
// BY THE GATE WAVES: the matrix waves issue nothing but MFMAs, fragment reads and the W_hh plane loads (round 3: every one of the
// four matrix waves split the same state fragment again -- 4x the VALU work, inside the MFMA stream).
// Decay: the gate waves store hd(s+1) = h(s) * exp(-lambda dt(s+1)) -- the product and the blend of position s+1 both want the
// decayed state (models/RRGCN.py:77-89) -- and keep it in registers for their own blend, so no fp32 state exists in LDS at all.
#pragma once

namespace temp {

#define CH2_NCX 8                        // chunks of 32 hidden columns: d <= 256
#define CH2_SPIN_MAX (1 << 18)           // x s_sleep(1): ~8 ms per wait before giving up
#define CH2_SYNC_BYTES 256               // hand-over words: [CH2_NCX][4] chunk progress of the gate waves + [4] positions of the matrix waves

__device__ int g_chain2_timeouts = 0;

struct Chain2Geom {
  int NT, NS, NC, lda;                   // forward: gate-column tiles (3d / 32), k-slabs of 16 (d), chunks, raw-product row stride
  int NTb, NSb, ldz, wlast;              // backward: state-column tiles, k'-slabs, raw-product row stride, width of the last chunk
};
__host__ __device__ inline Chain2Geom chain2_geom(int D) {
  Chain2Geom g;
  g.NT = (3 * D + 31) >> 5; g.NS = (D + 15) >> 4; g.NC = (D + 31) >> 5; g.lda = g.NT * 32 + 4;
  g.NTb = (D + 31) >> 5; g.wlast = D - 32 * (g.NC - 1);
  const int last = (3 * g.wlast + 15) >> 4;
  g.NSb = 6 * (g.NC - 1) + ((last + 1) & ~1);                 // slabs come in pairs (two plane sets in registers)
  g.ldz = g.NTb * 32 + 4;
  return g;
}
// k' (chunk-major order of the 3d gate columns) of gate g, column j of chunk c
__host__ __device__ inline int chain2_kprime(const Chain2Geom& g, int c, int gate, int j) {
  const int w = c < g.NC - 1 ? 32 : g.wlast;
  return 96 * c + gate * w + j;
}
inline size_t chain2_lds_fwd(int D, int ms) {
  const Chain2Geom g = chain2_geom(D);
  return (size_t)32 * g.lda * 4 + (size_t)g.NS * 3072 + (size_t)3 * D * 4 + (size_t)(2 * 32 + 1) * ms * 4 + CH2_SYNC_BYTES;
}
inline size_t chain2_lds_bwd(int D, int ms) {
  const Chain2Geom g = chain2_geom(D);
  return (size_t)g.NSb * 3072 + (size_t)32 * g.ldz * 4 + (size_t)(2 * 32 + 3) * ms * 4 + CH2_SYNC_BYTES;
}

// forward kernels are instantiated per TPW = ceil(ceil(3d / 32) / 4): the widest d of that class has this many chunks
__host__ __device__ constexpr int ch2_max_chunks(int tpw) { return tpw == 1 ? 2 : tpw == 2 ? 3 : tpw == 3 ? 4 : tpw == 4 ? 6 : tpw == 5 ? 7 : 8; }

// Hand-over words in LDS: every wave publishes its OWN progress (a plain store of "positions done" into its slot of the four
// words of a chunk / of the product), a consumer waits until all four slots reached its target.  (A shared counter is not
// enough: waves of a role are not in lockstep, and a wave that is a chunk ahead would stand in for one that is behind.)
__device__ __forceinline__ void ch2_wait4(const int* four, int target) {
  int it = 0;
  for (;;) {
    const int a = __hip_atomic_load(four, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int b = __hip_atomic_load(four + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int c = __hip_atomic_load(four + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int d = __hip_atomic_load(four + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (min(min(a, b), min(c, d)) >= target) break;
    __builtin_amdgcn_s_sleep(1);
    if (++it > CH2_SPIN_MAX) {
      if ((threadIdx.x & 63) == 0) atomicAdd(&g_chain2_timeouts, 1);
      break;
    }
  }
  asm volatile("" ::: "memory");
}
// this wave's LDS writes have landed, then one lane publishes the wave's progress
__device__ __forceinline__ void ch2_post(int* slot, int value) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// four consecutive elements -> the 8-byte halves of the three plane items they belong to
__device__ __forceinline__ void ch2_put_planes(bx_u32x4* planes, int kp, int track, const float4 v) {
  const int slab = kp >> 4, hh = (kp >> 3) & 1, half = (kp >> 2) & 1;
  unsigned h0, m0, l0, h1, m1, l1;
  bx_split_pair(v.x, v.y, h0, m0, l0);
  bx_split_pair(v.z, v.w, h1, m1, l1);
  uint2* dst = reinterpret_cast<uint2*>(planes + (size_t)(slab * 3) * 64 + hh * 32 + track) + half;
  dst[0] = make_uint2(h0, h1);
  dst[128] = make_uint2(m0, m1);
  dst[256] = make_uint2(l0, l1);
}


// Row streams the gate waves keep in flight for a WHOLE position (issued when a chunk is done, used one position later).  The
// compiler's wait-count pass merges pending loads at the loop header and then waits for everything outstanding -- including
// the prefetch issued a few instructions earlier (s_waitcnt vmcnt(0..2) in front of every chunk: an HBM round trip per chunk).
// So these loads are invisible to it (inline assembly) and waited for by count: VMEM operations of a wave complete in issue
// order on gfx950, and every chunk issues at least its own LOADS after the one being waited for (its stores only add to that).
#define CH2_LOAD_ASYNC(dst, ptr, byte_off) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(byte_off) : "memory")
// a per-position base pointer the compiler must keep as ONE register pair (chunk offsets are instruction immediates): without
// this it pre-computes every (plane, chunk) offset outside the position loop -- 35 pointer pairs, i.e. spills
#define CH2_OPAQUE(ptr) asm volatile("" : "+v"(ptr))
// (the pointers carry their address space in the type: a laundered generic pointer would turn every access into a flat_ one,
// which counts on BOTH wait counters)
typedef __attribute__((address_space(1))) f32x4 ch2_g4;         // global
typedef __attribute__((address_space(3))) f32x4 ch2_l4;         // LDS
__device__ __forceinline__ ch2_g4* ch2_gp(const float* p) { return (ch2_g4*)(p); }
__device__ __forceinline__ ch2_l4* ch2_lp(const float* p) { return (ch2_l4*)(p); }
__device__ __forceinline__ float4 ch2_ld(const ch2_l4* p, int i) { const f32x4 v = p[i]; return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float4 ch2_ld(const ch2_g4* p, int i) { const f32x4 v = p[i]; return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void ch2_st(ch2_g4* p, int i, const float4 v) { p[i] = f32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void ch2_st(ch2_l4* p, int i, const float4 v) { p[i] = f32x4{v.x, v.y, v.z, v.w}; }
// at most `n` vector-memory operations of this wave still in flight (n rounded down to a supported step)
__device__ __forceinline__ void ch2_vmwait(int n) {
  if (n >= 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
  else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (n >= 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if (n >= 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (n >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// the values below were written by ch2_load_async and are complete after the preceding ch2_vmwait: every use is ordered behind
#define CH2_LANDED3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c)::"memory")
#define CH2_LANDED5(a, b, c, d, e) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e)::"memory")
__device__ __forceinline__ float4 ch2_f4(const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }

// four consecutive elements back from their plane items: h + m + l is the exact fp32 value (gemm_bx.hpp)
__device__ __forceinline__ float4 ch2_get_planes(const bx_u32x4* planes, int kp, int track) {
  const int slab = kp >> 4, hh = (kp >> 3) & 1, half = (kp >> 2) & 1;
  const uint2* src = reinterpret_cast<const uint2*>(planes + (size_t)(slab * 3) * 64 + hh * 32 + track) + half;
  const uint2 h = src[0], m = src[128], l = src[256];
  auto lo = [](unsigned w) { return __uint_as_float(w << 16); };
  auto hi = [](unsigned w) { return __uint_as_float(w & 0xffff0000u); };
  return make_float4((lo(h.x) + lo(m.x)) + lo(l.x), (hi(h.x) + hi(m.x)) + hi(l.x), (lo(h.y) + lo(m.y)) + lo(l.y), (hi(h.y) + hi(m.y)) + hi(l.y));
}

// ---- W_hh -> both operand orders, one launch ----------------------------------------------------------------------------------
// forward  unit (slab s, tile t): 16-byte item of lane (li, hh), plane p = k 16 s + 8 hh .. +7 of gate column 32 t + li
//          (= k_bx_pack<1> of W_hh [3d][d]: the round-3 layout)
// backward unit (slab s, tile t): k' 16 s + 8 hh .. +7 (chunk-major gate columns) of state column 32 t + li; zero outside
__global__ void __launch_bounds__(256) k_chain2_pack(int D, const float* __restrict__ W, bx_u32x4* __restrict__ out) {
  const Chain2Geom g = chain2_geom(D);
  const int lane = threadIdx.x & 63, hh = lane >> 5, li = lane & 31;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nf = g.NS * g.NT, nb = g.NSb * g.NTb;
  if (unit >= nf + nb) return;
  float4 v0 = zero4(), v1 = zero4();
  if (unit < nf) {
    const int s = unit / g.NT, t = unit - s * g.NT;
    const int k = 16 * s + 8 * hh, n = 32 * t + li;
    if (n < 3 * D && k < D) { const float* p = W + (size_t)n * D + k; v0 = ld4(p); v1 = ld4(p + 4); }
  } else {
    const int u = unit - nf;
    const int s = u / g.NTb, t = u - s * g.NTb;
    const int kp = 16 * s + 8 * hh, n = 32 * t + li;
    int c = kp / 96;
    if (c > g.NC - 1) c = g.NC - 1;
    const int w = c < g.NC - 1 ? 32 : g.wlast;
    const int r = kp - 96 * c, gate = r / w, j = r - gate * w;
    if (gate < 3 && n < D) {
      const float* p = W + (size_t)(gate * D + 32 * c + j) * D + n;
      const size_t l = (size_t)D;
      v0 = make_float4(p[0], p[l], p[2 * l], p[3 * l]);
      v1 = make_float4(p[4 * l], p[5 * l], p[6 * l], p[7 * l]);
    }
  }
  bx_u32x4 H, Mi, L;
  bx_split8(v0, v1, H, Mi, L);
  bx_u32x4* d = out + (size_t)unit * 192 + lane;
  d[0] = H; d[64] = Mi; d[128] = L;
}

// ---- forward ------------------------------------------------------------------------------------------------------------------
// 512 threads: waves 0-3 matrix (one per SIMD, TPW tiles of 32 gate columns each), waves 4-7 gates.  A gate lane owns
// (track = item >> 3, four columns 4 (item & 7) of every chunk) for the whole panel: 8 lanes cover the 128 contiguous bytes
// of a track's chunk in every row stream.
template <int VARIANT, int TPW>
__global__ void __launch_bounds__(512) k_gru_chain_fwd2(ChainArgs a, const float* __restrict__ gi, float* __restrict__ H,
                                                        float* __restrict__ saved) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NCT = ch2_max_chunks(TPW);                             // chunks a kernel of this tile count can meet
  const int D = a.D;
  const int G = (VARIANT == TEMP_GRU_TORCH) ? 3 * D : D;
  const Chain2Geom g = chain2_geom(D);
  const int NT = g.NT, NS = g.NS, NC = g.NC, lda = g.lda;
  float* accb = lds;                                                   // [32][lda] raw products of the current position
  bx_u32x4* hp = reinterpret_cast<bx_u32x4*>(accb + CH_SLOTS * lda);   // [NS][3][64] decayed state, split, fragment order
  float* biasb = reinterpret_cast<float*>(hp + (size_t)NS * 192);      // [3d] b_hh
  int* tabb = reinterpret_cast<int*>(biasb + 3 * D);                   // [ms][32]
  float* decb = reinterpret_cast<float*>(tabb + CH_SLOTS * a.max_steps);
  int* flagb = reinterpret_cast<int*>(decb + CH_SLOTS * a.max_steps);
  int* cg = flagb + a.max_steps;                                       // [CH2_NCX][4] positions for which gate wave w has published chunk c
  int* cp = cg + 4 * CH2_NCX;                                          // [4] positions whose raw products matrix wave w has stored
  const int dbg = a.dbg;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const size_t plane = a.plane;

  {                       // one panel per workgroup (grid = n_panels): a panel loop makes every prefetch register loop-carried
    const int p = blockIdx.x;
    const int rnn_id = a.panel[4 * p], s0 = a.panel[4 * p + 1], ns = a.panel[4 * p + 2];
    const ChainRnn R = a.rnn[rnn_id];
    for (int i = tid; i < NS * 192; i += blockDim.x) hp[i] = bx_u32x4{0u, 0u, 0u, 0u};      // idle tracks / k padding: finite
    for (int i = tid; i < 3 * D; i += blockDim.x) biasb[i] = R.b_hh[i];
    for (int i = tid; i < ns * CH_SLOTS; i += blockDim.x) {
      const int e = a.rows[(size_t)s0 * CH_SLOTS + i];
      tabb[i] = e;
      decb[i] = e >= 0 ? expf(-a.dt[e & CH_ROW_MASK] * a.lambda) : 0.f;
    }
    if (tid < ns) flagb[tid] = a.sinfo[4 * (size_t)(s0 + tid)];
    if (tid < 4 * CH2_NCX + 4) cg[tid] = 0;
    __syncthreads();

    if (wave < 4) {
      // ------------------------------------------------------------------ matrix role
      if (!(dbg & 16)) __builtin_amdgcn_s_setprio(2);
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
      const bx_u32x4* wp = reinterpret_cast<const bx_u32x4*>(R.wf);
      bx_u32x4 wh[TPW], wm[TPW], wl[TPW];
      auto wload = [&](bx_u32x4 (&w)[TPW], int sl, int pl) {
        if (dbg & 8) return;
#pragma unroll
        for (int j = 0; j < TPW; ++j) w[j] = wp[((size_t)(sl * NT + tidx[j]) * 3 + pl) * 64 + lane];
      };
      // one slab (16 k) out of the plane registers; a plane is refilled with the next slab's as soon as its last round issued
      auto slab = [&](int sl, const bx_u32x4& FH, const bx_u32x4& FM, const bx_u32x4& FL) {
        const int sn = sl + 1 < NS ? sl + 1 : 0;                // past the end: slab 0 of the NEXT position
        const bx_bf16x8 ah = bx_frag(FH), am = bx_frag(FM), al = bx_frag(FL);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wl[j]), ah, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wload(wl, sn, 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wm[j]), am, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wm[j]), ah, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wload(wm, sn, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), al, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), am, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), ah, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wload(wh, sn, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      wload(wh, 0, 0); wload(wm, 0, 1); wload(wl, 0, 2);
      for (int s = 0; s < ns; ++s) {
        if (flagb[s] & 1) {
          for (int c = 0; c < NC; ++c) {
            if (!(dbg & 2)) ch2_wait4(cg + 4 * c, s);            // chunk c of hd(s) is in LDS
            const int sl0 = 2 * c;
            const bool two = sl0 + 1 < NS;
            const bx_u32x4* f0 = hp + (size_t)(sl0 * 3) * 64 + lane;
            const bx_u32x4 A0 = f0[0], A1 = f0[64], A2 = f0[128];
            bx_u32x4 B0 = A0, B1 = A1, B2 = A2;
            if (two) { B0 = f0[192]; B1 = f0[256]; B2 = f0[320]; }
            __builtin_amdgcn_sched_barrier(0);
            if (dbg & 4) continue;
            slab(sl0, A0, A1, A2);
            if (two) slab(sl0 + 1, B0, B1, B2);
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
        ch2_post(cp + wave, s + 1);                              // products of position s are in LDS (or were not needed)
      }
    } else {
      // ------------------------------------------------------------------ gate role
      if (dbg & 32) __builtin_amdgcn_s_setprio(3);
      const int item = (wave - 4) * 64 + lane, track = item >> 3, c4 = item & 7;
      f32x4 g0[NCT], g1[NCT], g2[NCT];
#pragma unroll
      for (int c = 0; c < NCT; ++c) { g0[c] = f32x4{0.f, 0.f, 0.f, 0.f}; g1[c] = g0[c]; g2[c] = g0[c]; }
      // loads a wave issues between a chunk's prefetch and its use
      const int n_after = (VARIANT == TEMP_GRU_TORCH) ? 3 * (NC - 1) : NC - 1;
      const int wl4 = (g.wlast >> 2);                             // float4 columns of the last chunk
      const bool lin = c4 < wl4;                                  // this lane's columns of the last chunk exist
      // input gates of a row for one position ahead: three base pointers (gate blocks r, z, n; type-1: n only), chunk offsets as
      // immediates; idle tracks read row 0, the columns past d of the last chunk gi[0..3] (valid, never used)
      auto prefetch_row = [&](int c, const float* b0, const float* b1, const float* b2) {
        const float* q0 = b0; const float* q1 = b1; const float* q2 = b2;
        if (c == NC - 1 && !lin) { q0 = gi - 32 * c; q1 = q0; q2 = q0; }
        if (VARIANT == TEMP_GRU_TORCH) { CH2_LOAD_ASYNC(g0[c], q0, 128 * c); CH2_LOAD_ASYNC(g1[c], q1, 128 * c); }
        CH2_LOAD_ASYNC(g2[c], q2, 128 * c);
      };
      auto gi_row = [&](int e) { return gi + (e >= 0 ? (size_t)(e & CH_ROW_MASK) * G : 0) + 4 * c4; };
      {
        const float* b0 = gi_row(tabb[track]);
        const float* b1 = b0 + ((VARIANT == TEMP_GRU_TORCH) ? D : 0);
        const float* b2 = b0 + ((VARIANT == TEMP_GRU_TORCH) ? 2 * D : 0);
        CH2_OPAQUE(b0); CH2_OPAQUE(b1); CH2_OPAQUE(b2);
#pragma unroll
        for (int c = 0; c < NCT; ++c) if (c < NC) prefetch_row(c, b0, b1, b2);
      }
      const ch2_l4* acc_r = ch2_lp(accb + (size_t)track * lda + 4 * c4);   // this lane's raw products: + 8 c (float4 units)
      const ch2_l4* acc_z = ch2_lp(accb + (size_t)track * lda + 4 * c4 + D);
      const ch2_l4* acc_n = ch2_lp(accb + (size_t)track * lda + 4 * c4 + 2 * D);
      const ch2_l4* bias_r = ch2_lp(biasb + 4 * c4);
      const ch2_l4* bias_z = ch2_lp(biasb + 4 * c4 + D);
      const ch2_l4* bias_n = ch2_lp(biasb + 4 * c4 + 2 * D);
      CH2_OPAQUE(acc_r); CH2_OPAQUE(acc_z); CH2_OPAQUE(acc_n); CH2_OPAQUE(bias_r); CH2_OPAQUE(bias_z); CH2_OPAQUE(bias_n);
      for (int s = 0; s < ns; ++s) {
        const int e = tabb[s * CH_SLOTS + track];
        const bool act = e >= 0, hp_ = act && (e & CH_HAS_PREV);
        const bool more = s + 1 < ns;
        const int en = more ? tabb[(s + 1) * CH_SLOTS + track] : -1;
        const float decn = (en >= 0 && (en & CH_HAS_PREV)) ? decb[(s + 1) * CH_SLOTS + track] : 0.f;
        const int flags = flagb[s];
        const size_t o = (size_t)(e & CH_ROW_MASK) * D + 4 * c4;
        ch2_g4* out_h = ch2_gp(H + o);
        ch2_g4* out_r = ch2_gp(saved + o);
        ch2_g4* out_z = ch2_gp(saved + plane + o);
        ch2_g4* out_n = ch2_gp(saved + 2 * plane + o);
        ch2_g4* out_hn = ch2_gp(saved + 3 * plane + o);
        ch2_g4* out_hd = ch2_gp(saved + 4 * plane + o);
        const float* b0 = gi_row(en);
        const float* b1 = b0 + ((VARIANT == TEMP_GRU_TORCH) ? D : 0);
        const float* b2 = b0 + ((VARIANT == TEMP_GRU_TORCH) ? 2 * D : 0);
        CH2_OPAQUE(out_h); CH2_OPAQUE(out_r); CH2_OPAQUE(out_z); CH2_OPAQUE(out_n); CH2_OPAQUE(out_hn); CH2_OPAQUE(out_hd);
        CH2_OPAQUE(b0); CH2_OPAQUE(b1); CH2_OPAQUE(b2);
        if (!(dbg & 2)) ch2_wait4(cp, s + 1);                    // raw products of position s are in LDS
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          if (c >= NC) break;
          const bool cin = c < NC - 1 || lin;
          ch2_vmwait(n_after);
          if (VARIANT == TEMP_GRU_TORCH) CH2_LANDED3(g0[c], g1[c], g2[c]); else asm volatile("" : "+v"(g2[c])::"memory");
          if (act && cin) {
            float4 ar = zero4(), az = zero4(), an = zero4(), hd = zero4();
            if (hp_) {                                  // (a track without a previous state contributed nothing it reads)
              ar = ch2_ld(acc_r, 8 * c); az = ch2_ld(acc_z, 8 * c); an = ch2_ld(acc_n, 8 * c);
              hd = ch2_get_planes(hp, 32 * c + 4 * c4, track);
            }
            const float4 bhr = ch2_ld(bias_r, 8 * c), bhz = ch2_ld(bias_z, 8 * c), bhn = ch2_ld(bias_n, 8 * c);
            float4 gi0 = zero4(), gi1 = zero4();
            if (VARIANT == TEMP_GRU_TORCH) { gi0 = ch2_f4(g0[c]); gi1 = ch2_f4(g1[c]); }
            const float4 gi2 = ch2_f4(g2[c]);
            float4 h4, rg4, zg4, ng4, hn4;
#define TEMP_CELL(cc)                                                                          \
            {                                                                                  \
              float xr = ar.cc, xz = az.cc;                                                    \
              if (VARIANT == TEMP_GRU_TORCH) { xr += gi0.cc; xz += gi1.cc; }                   \
              const float rg = gate_sigmoid(xr + bhr.cc);                                      \
              const float zg = gate_sigmoid(xz + bhz.cc);                                      \
              const float hn = an.cc + bhn.cc;                                                 \
              const float ng = gate_tanh(gi2.cc + rg * hn);                                    \
              h4.cc = (VARIANT == TEMP_GRU_TORCH) ? ((1.f - zg) * ng + zg * hd.cc) : (ng + zg * (hd.cc - ng)); \
              rg4.cc = rg; zg4.cc = zg; ng4.cc = ng; hn4.cc = hn;                              \
            }
            TEMP_CELL(x) TEMP_CELL(y) TEMP_CELL(z) TEMP_CELL(w)
#undef TEMP_CELL
            if (more) ch2_put_planes(hp, 32 * c + 4 * c4, track, scale4(h4, decn));   // the next position's decayed state
            if (!(dbg & 1)) {
              if (flags & 2) ch2_st(out_h, 8 * c, h4);
              ch2_st(out_r, 8 * c, rg4);
              ch2_st(out_z, 8 * c, zg4);
              ch2_st(out_n, 8 * c, ng4);
              ch2_st(out_hn, 8 * c, hn4);
              ch2_st(out_hd, 8 * c, hd);
            }
          }
          if (more) {
            ch2_post(cg + 4 * c + (wave - 4), s + 1);            // chunk c of hd(s + 1) is published
            prefetch_row(c, b0, b1, b2);                         // a whole position of flight time
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

// ---- backward -----------------------------------------------------------------------------------------------------------------
// 768 threads: waves 0-3 matrix (TPWB tiles of 32 state columns each), waves 4-11 gate gradients in two groups of four waves:
// group 0 takes the even chunks, group 1 the odd ones (a lane then keeps four chunks x five saved planes in flight: 80
// registers -- one group would need 140 at d = 200 -- and twice the waves hide the row streams).
template <int VARIANT, int TPWB>
__global__ void __launch_bounds__(768) k_gru_chain_bwd2(ChainArgs a, ChainUps ups, const float* __restrict__ saved,
                                                        float* __restrict__ dgi, float* __restrict__ dgh) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NCG = CH2_NCX / 2;                                     // chunks of a gate group
  const int D = a.D;
  const Chain2Geom g = chain2_geom(D);
  const int NC = g.NC, NTb = g.NTb, NSb = g.NSb, ldz = g.ldz;
  bx_u32x4* dp = reinterpret_cast<bx_u32x4*>(lds);                     // [NSb][3][64] dgh of the current position, split, k' order
  float* dpb = reinterpret_cast<float*>(dp + (size_t)NSb * 192);       // [32][ldz] dh * z of this position -> d_prev of this position
  int* tabb = reinterpret_cast<int*>(dpb + CH_SLOTS * ldz);            // [ms][32]
  float* decb = reinterpret_cast<float*>(tabb + CH_SLOTS * a.max_steps);
  int* flagb = reinterpret_cast<int*>(decb + CH_SLOTS * a.max_steps);  // [ms]
  int* upb = flagb + a.max_steps;                                      // [ms][2]
  int* cg = upb + 2 * a.max_steps;                                     // [CH2_NCX][4] positions for which wave w of the chunk's group has published it
  int* cp = cg + 4 * CH2_NCX;                                          // [4] positions whose d_prev matrix wave w has stored
  const int dbg = a.dbg;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const size_t plane = a.plane;

  {                       // one panel per workgroup (grid = n_panels): a panel loop makes every prefetch register loop-carried
    const int p = blockIdx.x;
    const int rnn_id = a.panel[4 * p], s0 = a.panel[4 * p + 1], ns = a.panel[4 * p + 2];
    const ChainRnn R = a.rnn[rnn_id];
    for (int i = tid; i < NSb * 192; i += blockDim.x) dp[i] = bx_u32x4{0u, 0u, 0u, 0u};
    if (tid < ns) { upb[2 * tid] = a.sinfo[4 * (size_t)(s0 + tid) + 1]; upb[2 * tid + 1] = a.sinfo[4 * (size_t)(s0 + tid) + 2]; }
    for (int i = tid; i < ns * CH_SLOTS; i += blockDim.x) {
      const int e = a.rows[(size_t)s0 * CH_SLOTS + i];
      tabb[i] = e;
      decb[i] = e >= 0 ? expf(-a.dt[e & CH_ROW_MASK] * a.lambda) : 0.f;
    }
    if (tid < ns) flagb[tid] = a.sinfo[4 * (size_t)(s0 + tid)];
    if (tid < 4 * CH2_NCX + 4) cg[tid] = 0;
    __syncthreads();

    if (wave < 4) {
      // ------------------------------------------------------------------ matrix role: raw d_prev = dgh . W_hh
      if (!(dbg & 16)) __builtin_amdgcn_s_setprio(2);
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
      // W_hh planes of TWO slabs in registers; the planes of slab sl + 2 replace those of slab sl as soon as their last round of
      // slab sl has issued (nine to eleven rounds of L2 latency cover)
      const bx_u32x4* wp = reinterpret_cast<const bx_u32x4*>(R.wb);
      bx_u32x4 wh0[TPWB], wm0[TPWB], wl0[TPWB], wh1[TPWB], wm1[TPWB], wl1[TPWB];
      auto wload = [&](bx_u32x4 (&w)[TPWB], int sl, int pl) {
        if (dbg & 8) return;
#pragma unroll
        for (int j = 0; j < TPWB; ++j) w[j] = wp[((size_t)(sl * NTb + tidx[j]) * 3 + pl) * 64 + lane];
      };
      auto slab = [&](bx_u32x4 (&wh)[TPWB], bx_u32x4 (&wm)[TPWB], bx_u32x4 (&wl)[TPWB], int sl2, const bx_u32x4& FH, const bx_u32x4& FM,
                      const bx_u32x4& FL) {
        const bx_bf16x8 ah = bx_frag(FH), am = bx_frag(FM), al = bx_frag(FL);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wl[j]), ah, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wload(wl, sl2, 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wm[j]), am, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wm[j]), ah, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wload(wm, sl2, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), al, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), am, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TPWB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx_frag(wh[j]), ah, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wload(wh, sl2, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      wload(wh0, 0, 0); wload(wm0, 0, 1); wload(wl0, 0, 2);
      wload(wh1, 1, 0); wload(wm1, 1, 1); wload(wl1, 1, 2);
      for (int s = ns - 1, i = 0; s >= 0; --s, ++i) {
        if (flagb[s] & 1) {
          for (int c = 0; c < NC; ++c) {
            if (!(dbg & 2)) ch2_wait4(cg + 4 * c, i + 1);        // the gate gradients of chunk c of position s are in LDS
            const int c0 = 6 * c, c1 = c < NC - 1 ? c0 + 6 : NSb;
            for (int sl = c0; sl < c1; sl += 2) {
              const int a2 = sl + 2 < NSb ? sl + 2 : 0, b2 = sl + 3 < NSb ? sl + 3 : 1;       // wrap: the NEXT position
              const bx_u32x4* f0 = dp + (size_t)(sl * 3) * 64 + lane;
              const bx_u32x4 A0 = f0[0], A1 = f0[64], A2 = f0[128], B0 = f0[192], B1 = f0[256], B2 = f0[320];
              __builtin_amdgcn_sched_barrier(0);
              if (dbg & 4) continue;
              slab(wh0, wm0, wl0, a2, A0, A1, A2);
              slab(wh1, wm1, wl1, b2, B0, B1, B2);
            }
          }
          const float dec = decb[s * CH_SLOTS + li];
#pragma unroll
          for (int j = 0; j < TPWB; ++j) {
            if (!tval[j]) continue;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              float* dst = dpb + (size_t)li * ldz + tidx[j] * 32 + 8 * qq + 4 * hh;
              const float4 gz = ld4(dst);                        // dh * z of this position, left here by the gate waves
              st4(dst, make_float4((acc[j][4 * qq] + gz.x) * dec, (acc[j][4 * qq + 1] + gz.y) * dec, (acc[j][4 * qq + 2] + gz.z) * dec,
                                   (acc[j][4 * qq + 3] + gz.w) * dec));
              acc[j][4 * qq] = 0.f; acc[j][4 * qq + 1] = 0.f; acc[j][4 * qq + 2] = 0.f; acc[j][4 * qq + 3] = 0.f;
            }
          }
        }
        ch2_post(cp + wave, i + 1);                              // d_prev of position s is in LDS (or nobody consumes it)
      }
    } else {
      // ------------------------------------------------------------------ gate role: gate gradients
      if (dbg & 32) __builtin_amdgcn_s_setprio(3);
      const int grp = (wave - 4) >> 2;
      const int item = ((wave - 4) & 3) * 64 + lane, track = item >> 3, c4 = item & 7;
      f32x4 sr[NCG], sz[NCG], sn[NCG], shn[NCG], shd[NCG];
#pragma unroll
      for (int q = 0; q < NCG; ++q) { sr[q] = f32x4{0.f, 0.f, 0.f, 0.f}; sz[q] = sr[q]; sn[q] = sr[q]; shn[q] = sr[q]; shd[q] = sr[q]; }
      const int my_chunks = (NC - grp + 1) >> 1;                  // chunks grp, grp + 2, ... of this wave
      const int n_after = 5 * (my_chunks - 1);                    // loads the wave issues between a chunk's prefetch and its use
      const int wl4 = (g.wlast >> 2);
      const bool lin = c4 < wl4;
      // the five saved planes of a row for one position ahead (chunk offsets as immediates; idle tracks read row 0, the columns
      // past d of the last chunk saved[0..3])
      auto prefetch_row = [&](int q, const float* p0) {
        const int c = grp + 2 * q;
        const float* a0 = p0;
        size_t pl = plane;
        if (c == NC - 1 && !lin) { a0 = saved - 32 * c; pl = 0; }
        CH2_OPAQUE(a0);                                           // the four plane pointers are built here, per chunk, and die here
        const float* a1 = a0 + pl; const float* a2 = a1 + pl; const float* a3 = a2 + pl; const float* a4 = a3 + pl;
        // q-th chunk of this wave: byte offset 128 (grp + 2 q) = 256 q on top of the group's own 128 grp (folded into the pointer)
        CH2_LOAD_ASYNC(sr[q], a0, 256 * q); CH2_LOAD_ASYNC(sz[q], a1, 256 * q); CH2_LOAD_ASYNC(sn[q], a2, 256 * q);
        CH2_LOAD_ASYNC(shn[q], a3, 256 * q); CH2_LOAD_ASYNC(shd[q], a4, 256 * q);
      };
      auto saved_row = [&](int e) { return saved + (e >= 0 ? (size_t)(e & CH_ROW_MASK) * D : 0) + 4 * c4 + 32 * grp; };
      {
        const float* p0 = saved_row(tabb[(ns - 1) * CH_SLOTS + track]);
        CH2_OPAQUE(p0);
#pragma unroll
        for (int q = 0; q < NCG; ++q) {
          if (grp + 2 * q < NC) prefetch_row(q, p0);
        }
      }
      ch2_l4* dpr = ch2_lp(dpb + (size_t)track * ldz + 4 * c4 + 32 * grp);    // d_prev(s + 1) in, dh * z (s) out: + 16 q
      CH2_OPAQUE(dpr);
      for (int s = ns - 1, i = 0; s >= 0; --s, ++i) {
        const int e = tabb[s * CH_SLOTS + track];
        const bool act = e >= 0;
        const size_t row = (size_t)(e & CH_ROW_MASK);
        const int eu = s + 1 < ns ? tabb[(s + 1) * CH_SLOTS + track] : -1;
        const bool nxt = eu >= 0 && (eu & CH_HAS_PREV);           // position s + 1 consumed this row's state
        const float* p0 = saved_row(s > 0 ? tabb[(s - 1) * CH_SLOTS + track] : -1);
        // upstream gradient of the step's rows (only the positions whose states are consumed outside the chain have one)
        const int up_sel = upb[2 * s], up_row0 = upb[2 * s + 1];
        const bool has_up = up_sel >= 0;                          // (uniform)
        const size_t go = ((VARIANT == TEMP_GRU_TORCH) ? row * 3 * D : row * D) + 4 * c4 + 32 * grp;
        const size_t ho = row * 3 * D + 4 * c4 + 32 * grp;
        const float* gi_b = dgi + go;
        const float* gh_b = dgh + ho;
        CH2_OPAQUE(p0); CH2_OPAQUE(gi_b); CH2_OPAQUE(gh_b);
        // upstream rows: issued now, youngest loads of the wave until the chunk loop starts
        f32x4 upv[NCG];
#pragma unroll
        for (int q = 0; q < NCG; ++q) upv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_up) {
          const float* upp = ups.p[up_sel] + (act ? (row - (size_t)up_row0) * D : 0) + 4 * c4 + 32 * grp;
          CH2_OPAQUE(upp);
#pragma unroll
          for (int q = 0; q < NCG; ++q) {
            const int c = grp + 2 * q;
            if (c < NC) { const float* u = (c == NC - 1 && !lin) ? ups.p[up_sel] - 32 * c : upp; CH2_LOAD_ASYNC(upv[q], u, 256 * q); }
          }
        }
        if (!(dbg & 2)) ch2_wait4(cp, i);                        // d_prev of position s + 1 is in LDS
#pragma unroll
        for (int q = 0; q < NCG; ++q) {
          const int c = grp + 2 * q;
          if (c >= NC) break;
          const bool cin = c < NC - 1 || lin;
          const int w = c < NC - 1 ? 32 : g.wlast;
          // the chunk's upstream row (if any) is younger than its saved planes: behind it, the rest of the up loads and the
          // prefetches of the chunks already done
          if (has_up) { ch2_vmwait(my_chunks - 1 - q + (s > 0 ? 5 * q : 0)); asm volatile("" : "+v"(upv[q])::"memory"); }
          else ch2_vmwait(n_after);
          CH2_LANDED5(sr[q], sz[q], sn[q], shn[q], shd[q]);
          float4 gd = ch2_f4(upv[q]);
          if (act && cin) {
            if (nxt) gd = add4(gd, ch2_ld(dpr, 16 * q));          // d_prev = (dgh . W_hh + dh z) * decay (models/RRGCN.py:83, backward)
            const float4 rg = ch2_f4(sr[q]), zg = ch2_f4(sz[q]), ng = ch2_f4(sn[q]), hn = ch2_f4(shn[q]), hd = ch2_f4(shd[q]);
            float4 dr_pre, dz_pre, dn_pre, dhn, gz;
#define TEMP_GATE(cc)                                            \
            {                                                    \
              const float dn = gd.cc * (1.f - zg.cc);            \
              const float dz = gd.cc * (hd.cc - ng.cc);          \
              dn_pre.cc = dn * (1.f - ng.cc * ng.cc);            \
              dr_pre.cc = dn_pre.cc * hn.cc * rg.cc * (1.f - rg.cc); \
              dz_pre.cc = dz * zg.cc * (1.f - zg.cc);            \
              dhn.cc = dn_pre.cc * rg.cc;                        \
              gz.cc = gd.cc * zg.cc;                             \
            }
            TEMP_GATE(x) TEMP_GATE(y) TEMP_GATE(z) TEMP_GATE(w)
#undef TEMP_GATE
            ch2_st(dpr, 16 * q, gz);                             // the matrix waves add it to this position's product
            const int kp = 96 * c + 4 * c4;
            ch2_put_planes(dp, kp, track, dr_pre);
            ch2_put_planes(dp, kp + w, track, dz_pre);
            ch2_put_planes(dp, kp + 2 * w, track, dhn);
            if (!(dbg & 1)) {                                    // (the gate-block pointers are built here, per chunk, and die here)
              const float* gq = gi_b; const float* hq = gh_b;
              CH2_OPAQUE(gq); CH2_OPAQUE(hq);
              if (VARIANT == TEMP_GRU_TORCH) { ch2_st(ch2_gp(gq), 16 * q, dr_pre); ch2_st(ch2_gp(gq + D), 16 * q, dz_pre); ch2_st(ch2_gp(gq + 2 * D), 16 * q, dn_pre); }
              else ch2_st(ch2_gp(gq), 16 * q, dn_pre);
              ch2_st(ch2_gp(hq), 16 * q, dr_pre); ch2_st(ch2_gp(hq + D), 16 * q, dz_pre); ch2_st(ch2_gp(hq + 2 * D), 16 * q, dhn);
            }
          }
          ch2_post(cg + 4 * c + ((wave - 4) & 3), i + 1);        // chunk c of dgh(s) is published
          if (s > 0) prefetch_row(q, p0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

}  // namespace temp
