// Host-side planner (see include/temp_amd_host.h).  Plain C++17; built into temp_amd/libtemp_host.so by temp_amd/build.py.
#include "temp_amd_host.h"
#include <algorithm>
#include <cstddef>
#include <vector>

extern "C" {

int temp_host_abi_version(void) { return 2; }

}  // extern "C"

// Edge ids sorted by (seg, b) when sort_b (else by seg alone), ties in input order: two stable counting sorts.  The by-destination
// and by-source views list a segment's edges in RELATION order (b = relation there), so the chunks of a hub are runs of one
// relation: the edge kernels then read a run's block weights once (rgcn_kernels.hip, relation runs).  Any order inside a
// segment is a valid view; this one is fixed by the input alone, so results stay deterministic.
static bool sorted_edge_order(int64_t E, const int64_t* seg, const int64_t* b, int64_t n_seg, bool sort_b, std::vector<int64_t>& ptr,
                              std::vector<int64_t>& order) {
  ptr.assign((size_t)n_seg + 1, 0);
  for (int64_t e = 0; e < E; ++e) {
    if (seg[e] < 0 || seg[e] >= n_seg) return false;
    ++ptr[(size_t)seg[e] + 1];
  }
  for (int64_t s = 0; s < n_seg; ++s) ptr[(size_t)s + 1] += ptr[(size_t)s];
  order.resize((size_t)E);
  std::vector<int64_t> first;
  if (sort_b && E > 0) {
    int64_t nb = 0;
    for (int64_t e = 0; e < E; ++e) { if (b[e] < 0) return false; nb = b[e] + 1 > nb ? b[e] + 1 : nb; }
    std::vector<int64_t> bp((size_t)nb + 1, 0);
    for (int64_t e = 0; e < E; ++e) ++bp[(size_t)b[e] + 1];
    for (int64_t k = 0; k < nb; ++k) bp[(size_t)k + 1] += bp[(size_t)k];
    first.resize((size_t)E);
    for (int64_t e = 0; e < E; ++e) first[(size_t)bp[(size_t)b[e]]++] = e;
  }
  std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
  for (int64_t i = 0; i < E; ++i) {
    const int64_t e = first.empty() ? i : first[(size_t)i];
    order[(size_t)cur[(size_t)seg[e]]++] = e;
  }
  return true;
}

extern "C" {

int temp_host_build_view(int64_t E, const int64_t* seg, const int64_t* a, const int64_t* b, int64_t n_seg, int64_t chunk, int sort_b,
                         int64_t* order, int32_t* a_out, int32_t* b_out,
                         int32_t* chunk_seg, int32_t* chunk_beg, int32_t* chunk_end, int32_t* chunk_slot,
                         int32_t* fix_seg, int32_t* fix_slot, int32_t* fix_cnt, int64_t* counts) {
  if (E < 0 || n_seg < 0 || chunk <= 0 || !counts || (E > 0 && (!seg || !a || !b || !order || !a_out || !b_out))) return 1;
  std::vector<int64_t> ptr, ord;
  if (!sorted_edge_order(E, seg, b, n_seg, sort_b != 0, ptr, ord)) return 2;
  for (int64_t i = 0; i < E; ++i) order[i] = ord[(size_t)i];
  for (int64_t i = 0; i < E; ++i) { a_out[i] = (int32_t)a[order[i]]; b_out[i] = (int32_t)b[order[i]]; }
  int64_t n_chunks = 0, n_partial = 0, n_fix = 0;
  for (int64_t s = 0; s < n_seg; ++s) {
    const int64_t beg = ptr[(size_t)s], end = ptr[(size_t)s + 1];
    const int64_t nch = (end - beg + chunk - 1) / chunk;
    if (nch > 1) { fix_seg[n_fix] = (int32_t)s; fix_slot[n_fix] = (int32_t)n_partial; fix_cnt[n_fix] = (int32_t)nch; ++n_fix; }
    for (int64_t k = 0; k < nch; ++k) {
      chunk_seg[n_chunks] = (int32_t)s;
      chunk_beg[n_chunks] = (int32_t)(beg + k * chunk);
      chunk_end[n_chunks] = (int32_t)((beg + (k + 1) * chunk < end) ? beg + (k + 1) * chunk : end);
      chunk_slot[n_chunks] = nch > 1 ? (int32_t)n_partial++ : -1;
      ++n_chunks;
    }
  }
  counts[0] = n_chunks; counts[1] = n_partial; counts[2] = n_fix;
  return 0;
}

int temp_host_chain_plan(int bsz, int64_t num_ents, int n_steps, const int32_t* pos, const int32_t* n_win,
                         const int64_t* const* gids, const int64_t* gid_n,
                         int32_t* prev_idx, int32_t* next_idx, float* dt, int64_t* row_of, float* last) {
  if (bsz < 0 || num_ents < 0 || n_steps < 0 || !row_of || !last || (n_steps > 0 && (!pos || !n_win || !gids || !gid_n))) return 1;
  const size_t total = (size_t)bsz * (size_t)num_ents;
  for (size_t i = 0; i < total; ++i) { row_of[i] = -1; last[i] = 0.f; }
  int64_t out = 0, prev_out = 0;                       // first output row of this step / of the previous executed step
  int prev_step = -1;
  for (int s = 0; s < n_steps; ++s) {
    const int64_t step_out = out;
    const int nw = n_win[s];
    if (nw < 0 || nw > bsz) return 2;
    const float p = (float)pos[s];
    // read the maps left by the previous executed step
    for (int j = 0; j < nw; ++j) {
      const int64_t* g = gids[(size_t)s * bsz + j];
      const int64_t n = gid_n[(size_t)s * bsz + j];
      int64_t* ro = row_of + (size_t)j * num_ents;
      float* la = last + (size_t)j * num_ents;
      for (int64_t i = 0; i < n; ++i) {
        const int64_t e = g[i];
        if (e < 0 || e >= num_ents) return 3;
        prev_idx[out + i] = (int32_t)ro[e];
        next_idx[out + i] = -1;
        if (ro[e] >= 0) next_idx[prev_out + ro[e]] = (int32_t)(out + i - step_out);
        dt[out + i] = p - la[e];
      }
      out += n;
    }
    prev_out = step_out;
    // the history holds ONLY this step's nodes: forget the previous step's rows, then record this step's
    if (prev_step >= 0) {
      const int pw = n_win[prev_step];
      for (int j = 0; j < pw; ++j) {
        const int64_t* g = gids[(size_t)prev_step * bsz + j];
        const int64_t n = gid_n[(size_t)prev_step * bsz + j];
        int64_t* ro = row_of + (size_t)j * num_ents;
        for (int64_t i = 0; i < n; ++i) ro[g[i]] = -1;
      }
    }
    int64_t row = 0;
    for (int j = 0; j < nw; ++j) {
      const int64_t* g = gids[(size_t)s * bsz + j];
      const int64_t n = gid_n[(size_t)s * bsz + j];
      int64_t* ro = row_of + (size_t)j * num_ents;
      float* la = last + (size_t)j * num_ents;
      for (int64_t i = 0; i < n; ++i) { ro[g[i]] = row + i; la[g[i]] = p; }
      row += n;
    }
    prev_step = s;
  }
  return 0;
}

int temp_host_plan_loss(int n_graphs, const int64_t* graph_ptrs, const int64_t* const* idx, const int64_t* n_pos, const int64_t* row_offset,
                        int pad4, int64_t R, int32_t* packed, float* weights, int64_t* triples) {
  if (n_graphs < 0 || R < 0 || (n_graphs > 0 && (!graph_ptrs || !idx || !n_pos || !row_offset)) || (R > 0 && (!packed || !weights || !triples))) return 1;
  int32_t* known = packed;
  int32_t* rel_o = packed + R;
  int32_t* tail = packed + 2 * R;
  int32_t* truth = packed + 3 * R;
  int32_t* lo = packed + 4 * R;
  int32_t* hi = packed + 5 * R;
  int64_t row = 0, trow = 0;
  for (int g = 0; g < n_graphs; ++g) {
    const int64_t P = n_pos[g];
    const int64_t pad = (pad4 && P > 0) ? ((4 - (2 * P) % 4) % 4) : 0;
    if (P < 0 || row + 2 * P + pad > R) return 2;
    if (P == 0) continue;
    const int64_t* gp = graph_ptrs + (size_t)g * 8;
    const int64_t* src = (const int64_t*)gp[0];
    const int64_t* rel = (const int64_t*)gp[1];
    const int64_t* dst = (const int64_t*)gp[2];
    const int64_t* gid = (const int64_t*)gp[3];
    const int32_t* tlo = (const int32_t*)gp[4];
    const int32_t* thi = (const int32_t*)gp[5];
    const int32_t* hlo = (const int32_t*)gp[6];
    const int32_t* hhi = (const int32_t*)gp[7];
    const float w = 1.0f / (float)P;
    const int64_t off = row_offset[g];
    for (int64_t i = 0; i < P; ++i) {
      const int64_t e = idx[g][i];
      const int64_t s = src[e], r = rel[e], d = dst[e];
      triples[3 * (trow + i)] = s; triples[3 * (trow + i) + 1] = r; triples[3 * (trow + i) + 2] = d;
      const int64_t a = row + i, b = row + P + i;
      known[a] = (int32_t)(s + off); rel_o[a] = (int32_t)r; tail[a] = 1; truth[a] = (int32_t)gid[d]; lo[a] = tlo[e]; hi[a] = thi[e]; weights[a] = w;
      known[b] = (int32_t)(d + off); rel_o[b] = (int32_t)r; tail[b] = 0; truth[b] = (int32_t)gid[s]; lo[b] = hlo[e]; hi[b] = hhi[e]; weights[b] = w;
    }
    row += 2 * P;
    for (int64_t i = 0; i < pad; ++i, ++row) {
      known[row] = (int32_t)off; rel_o[row] = 0; tail[row] = 0; truth[row] = 0; lo[row] = 0; hi[row] = 0; weights[row] = 0.f;
    }
    trow += P;
  }
  return row == R ? 0 : 3;
}

// one view written compactly at `out`; returns entries written (or -1), fills sizes[9], *n_partial, optionally per-segment
// chunk counts and per-chunk ranks (by-relation view)
static int64_t pack_view(int64_t E, const int64_t* seg, const int64_t* a, const int64_t* b, int64_t n_seg, int64_t chunk, bool sort_b, int32_t* out,
                         int64_t* sizes, int64_t* n_partial, int32_t* seg_count /* nullable [n_seg] */, int64_t* seg_chunks /* nullable [n_seg] */,
                         std::vector<int32_t>* rank /* nullable */) {
  std::vector<int64_t> ptr, ord;
  if (!sorted_edge_order(E, seg, b, n_seg, sort_b, ptr, ord)) return -1;
  if (seg_count) for (int64_t s = 0; s < n_seg; ++s) seg_count[s] = (int32_t)(ptr[(size_t)s + 1] - ptr[(size_t)s]);
  int64_t n_chunks = 0, n_fix = 0;
  for (int64_t s = 0; s < n_seg; ++s) {
    const int64_t nch = (ptr[(size_t)s + 1] - ptr[(size_t)s] + chunk - 1) / chunk;
    n_chunks += nch;
    n_fix += nch > 1;
    if (seg_chunks) seg_chunks[s] = nch;
  }
  int32_t* a_o = out;
  int32_t* b_o = a_o + E;
  int32_t* c_seg = b_o + E;
  int32_t* c_beg = c_seg + n_chunks;
  int32_t* c_end = c_beg + n_chunks;
  int32_t* c_slot = c_end + n_chunks;
  int32_t* f_seg = c_slot + n_chunks;
  int32_t* f_slot = f_seg + n_fix;
  int32_t* f_cnt = f_slot + n_fix;
  for (int64_t p = 0; p < E; ++p) { const int64_t e = ord[(size_t)p]; a_o[p] = (int32_t)a[e]; b_o[p] = (int32_t)b[e]; }
  if (rank) rank->resize((size_t)n_chunks);
  int64_t c = 0, part = 0, f = 0;
  for (int64_t s = 0; s < n_seg; ++s) {
    const int64_t beg = ptr[(size_t)s], end = ptr[(size_t)s + 1];
    const int64_t nch = (end - beg + chunk - 1) / chunk;
    if (nch > 1) { f_seg[f] = (int32_t)s; f_slot[f] = (int32_t)part; f_cnt[f] = (int32_t)nch; ++f; }
    for (int64_t k = 0; k < nch; ++k, ++c) {
      c_seg[c] = (int32_t)s;
      c_beg[c] = (int32_t)(beg + k * chunk);
      c_end[c] = (int32_t)((beg + (k + 1) * chunk < end) ? beg + (k + 1) * chunk : end);
      c_slot[c] = nch > 1 ? (int32_t)part++ : -1;
      if (rank) (*rank)[(size_t)c] = (int32_t)k;
    }
  }
  sizes[0] = E; sizes[1] = E; sizes[2] = sizes[3] = sizes[4] = sizes[5] = n_chunks; sizes[6] = sizes[7] = sizes[8] = n_fix;
  *n_partial = part;
  return 2 * E + 4 * n_chunks + 3 * n_fix;
}

int64_t temp_host_snapshot_pack(int64_t n, int64_t E, const int64_t* src, const int64_t* dst, const int64_t* rel, const float* nnorm,
                                int64_t n_rel_rows, int64_t chunk, int64_t chunk_rel,
                                int32_t* packed, int64_t* sizes, int64_t* n_partial, int64_t* rel_chunks) {
  if (n < 0 || E < 0 || n_rel_rows < 0 || chunk <= 0 || chunk_rel <= 0 || !packed || !sizes || !n_partial || (n_rel_rows > 0 && !rel_chunks) ||
      (E > 0 && (!src || !dst || !rel)) || (n > 0 && !nnorm)) return -1;
  std::vector<int32_t> in_deg((size_t)n), out_deg((size_t)n), rank;
  int64_t off = 0;
  int64_t w = pack_view(E, dst, src, rel, n, chunk, true, packed + off, sizes, n_partial, in_deg.data(), nullptr, nullptr);
  if (w < 0) return -1;
  off += w;
  w = pack_view(E, src, dst, rel, n, chunk, true, packed + off, sizes + 9, n_partial + 1, out_deg.data(), nullptr, nullptr);
  if (w < 0) return -1;
  off += w;
  w = pack_view(E, rel, src, dst, n_rel_rows, chunk_rel, true, packed + off, sizes + 18, n_partial + 2, nullptr, rel_chunks, &rank);
  if (w < 0) return -1;
  off += w;
  for (size_t i = 0; i < rank.size(); ++i) packed[off + (int64_t)i] = rank[i];
  sizes[27] = (int64_t)rank.size();
  off += (int64_t)rank.size();
  for (int64_t i = 0; i < n; ++i) packed[off + i] = in_deg[(size_t)i];
  off += n;
  for (int64_t i = 0; i < n; ++i) packed[off + i] = out_deg[(size_t)i];
  off += n;
  const int32_t* bits = reinterpret_cast<const int32_t*>(nnorm);
  for (int64_t i = 0; i < n; ++i) packed[off + i] = bits[i];
  off += n;
  sizes[28] = sizes[29] = sizes[30] = n;
  return off;
}

int64_t temp_host_union_plan(int64_t M, int64_t R, const int64_t* meta, const int64_t* node_off, const int64_t* edge_off,
                             int64_t piece, int32_t* ctl, int64_t ctl_cap, int64_t* summary) {
  if (M < 0 || R < 0 || piece <= 0 || !ctl || !summary || (M > 0 && (!meta || !node_off || !edge_off))) return -1;
  const int64_t W = 66 + R;
  enum { ZERO, NODE, EDGE, PDST, PSRC, MEMB };
  struct Spec { int col, add, mode, aux; };
  static const Spec spec[30] = {
      {28, ZERO, 0, -1}, {29, ZERO, 0, -1}, {30, ZERO, 0, -1},
      {0, NODE, 0, -1}, {1, ZERO, 0, -1}, {2, NODE, 0, -1}, {3, EDGE, 0, -1}, {4, EDGE, 0, -1}, {5, PDST, 1, -1}, {6, NODE, 0, -1}, {7, PDST, 0, -1}, {8, ZERO, 0, -1},
      {9, NODE, 0, -1}, {10, ZERO, 0, -1}, {11, NODE, 0, -1}, {12, EDGE, 0, -1}, {13, EDGE, 0, -1}, {14, PSRC, 1, -1}, {15, NODE, 0, -1}, {16, PSRC, 0, -1}, {17, ZERO, 0, -1},
      {18, NODE, 0, -1}, {19, NODE, 0, -1}, {20, ZERO, 0, -1}, {21, EDGE, 0, -1}, {22, EDGE, 0, -1}, {27, MEMB, 2, 20}};
  const int n_out = 27;                                  // 3 + 9 + 9 + 6
  auto size_of = [&](int64_t m, int c) { return meta[m * W + c]; };
  auto off_of = [&](int64_t m, int c) { return meta[m * W + 31 + c]; };
  auto ptr_of = [&](int64_t m) { return meta[m * W + 62]; };
  // partial-slot offsets of the node views, per-relation chunk totals
  std::vector<int64_t> p_dst((size_t)M), p_src((size_t)M), per_rel((size_t)R, 0);
  int64_t acc_d = 0, acc_s = 0;
  for (int64_t m = 0; m < M; ++m) {
    p_dst[(size_t)m] = acc_d; acc_d += meta[m * W + 63];
    p_src[(size_t)m] = acc_s; acc_s += meta[m * W + 64];
    for (int64_t r = 0; r < R; ++r) per_rel[(size_t)r] += meta[m * W + 66 + r];
  }
  int64_t n_fix = 0, rel_partial = 0;
  std::vector<int64_t> base((size_t)R, -1);
  for (int64_t r = 0; r < R; ++r)
    if (per_rel[(size_t)r] > 1) { base[(size_t)r] = rel_partial; rel_partial += per_rel[(size_t)r]; ++n_fix; }
  // sizes of the sections
  int64_t n_desc = 0, n_pieces = 0;
  int64_t totals[30] = {0};
  for (int o = 0; o < n_out; ++o)
    for (int64_t m = 0; m < M; ++m) {
      const int64_t len = size_of(m, spec[o].col);
      totals[o] += len;
      if (len > 0) { ++n_desc; n_pieces += (len + piece - 1) / piece; }
    }
  const int64_t need = 8 * n_desc + 2 * n_pieces + M * R + 3 * n_fix;
  if (need > ctl_cap) return -1;
  int32_t* desc = ctl;
  int32_t* pd = ctl + 8 * n_desc;
  int32_t* ps = pd + n_pieces;
  int32_t* table = ps + n_pieces;
  int32_t* f_seg = table + M * R;
  int32_t* f_slot = f_seg + n_fix;
  int32_t* f_cnt = f_slot + n_fix;
  int64_t out_base = 0, d = 0, pc = 0;
  for (int o = 0; o < n_out; ++o) {
    summary[4 + o] = out_base;
    int64_t dst = out_base;
    for (int64_t m = 0; m < M; ++m) {
      const int64_t len = size_of(m, spec[o].col);
      if (len > 0) {
        int64_t add = 0;
        switch (spec[o].add) {
          case NODE: add = node_off[m]; break;
          case EDGE: add = edge_off[m]; break;
          case PDST: add = p_dst[(size_t)m]; break;
          case PSRC: add = p_src[(size_t)m]; break;
          case MEMB: add = m * R; break;
          default: break;
        }
        const int64_t src = ptr_of(m) + 4 * off_of(m, spec[o].col);
        const int64_t aux = ptr_of(m) + 4 * off_of(m, spec[o].aux < 0 ? 0 : spec[o].aux);
        int32_t* rec = desc + 8 * d;
        rec[0] = (int32_t)(uint32_t)(src & 0xffffffffll); rec[1] = (int32_t)(src >> 32);
        rec[2] = (int32_t)(uint32_t)(aux & 0xffffffffll); rec[3] = (int32_t)(aux >> 32);
        rec[4] = (int32_t)dst; rec[5] = (int32_t)len; rec[6] = (int32_t)add; rec[7] = spec[o].mode;
        for (int64_t st = 0; st < len; st += piece, ++pc) { pd[pc] = (int32_t)d; ps[pc] = (int32_t)st; }
        ++d;
      }
      dst += len;
    }
    out_base += totals[o];
  }
  summary[4 + n_out] = out_base;
  // slot table of the by-relation view: member m's chunks of relation r start at base[r] + (chunks of r in earlier members)
  {
    std::vector<int64_t> run((size_t)R, 0);
    for (int64_t m = 0; m < M; ++m)
      for (int64_t r = 0; r < R; ++r) {
        table[m * R + r] = per_rel[(size_t)r] > 1 ? (int32_t)(base[(size_t)r] + run[(size_t)r]) : -1;
        run[(size_t)r] += meta[m * W + 66 + r];
      }
  }
  for (int64_t r = 0, f = 0; r < R; ++r)
    if (per_rel[(size_t)r] > 1) { f_seg[f] = (int32_t)r; f_slot[f] = (int32_t)base[(size_t)r]; f_cnt[f] = (int32_t)per_rel[(size_t)r]; ++f; }
  summary[0] = n_desc; summary[1] = n_pieces; summary[2] = n_fix; summary[3] = out_base;
  for (int o = 0; o < n_out; ++o) summary[4 + 31 + o] = totals[o];
  summary[4 + 31 + 30] = acc_d; summary[4 + 31 + 31] = acc_s; summary[4 + 31 + 32] = rel_partial;
  return need;
}

static inline uint64_t splitmix64_next(uint64_t& state) {
  uint64_t x = (state += 0x9E3779B97F4A7C15ull);
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

int temp_host_sample_subset(int64_t n, int64_t k, uint64_t seed, int64_t* out) {
  if (n < 0 || k < 0 || k > n || (k > 0 && !out)) return 1;
  std::vector<int64_t> pool((size_t)n);
  for (int64_t i = 0; i < n; ++i) pool[(size_t)i] = i;
  uint64_t state = seed;
  for (int64_t i = 0; i < k; ++i) {
    // unbiased enough for sampling edges: 64-bit draw reduced by multiply-shift over the n - i remaining slots
    const uint64_t r = splitmix64_next(state);
    const uint64_t span = (uint64_t)(n - i);
    const int64_t j = i + (int64_t)(((unsigned __int128)r * span) >> 64);
    const int64_t t = pool[(size_t)i]; pool[(size_t)i] = pool[(size_t)j]; pool[(size_t)j] = t;
    out[i] = pool[(size_t)i];
  }
  return 0;
}

int64_t temp_host_gather_inverse(int64_t n, const int64_t* idx, int64_t n_rows, int32_t* seg_ptr, int32_t* order) {
  if (n < 0 || n_rows < 0 || !seg_ptr || (n > 0 && (!idx || !order))) return -1;
  for (int64_t r = 0; r <= n_rows; ++r) seg_ptr[r] = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (idx[i] >= n_rows) return -1;
    if (idx[i] >= 0) ++seg_ptr[idx[i] + 1];
  }
  for (int64_t r = 0; r < n_rows; ++r) seg_ptr[r + 1] += seg_ptr[r];
  std::vector<int32_t> cur(seg_ptr, seg_ptr + n_rows);
  for (int64_t i = 0; i < n; ++i)
    if (idx[i] >= 0) order[cur[(size_t)idx[i]]++] = (int32_t)i;
  return seg_ptr[n_rows];
}

int64_t temp_host_unique_labels(int64_t n, const int64_t* labels, int64_t n_labels, int32_t* first, int32_t* inv) {
  if (n < 0 || n_labels < 0 || (n > 0 && (!labels || !first || !inv))) return -1;
  std::vector<int32_t> slot((size_t)n_labels, -1);           // first position of a label, then its rank
  for (int64_t i = 0; i < n; ++i) {
    if (labels[i] < 0 || labels[i] >= n_labels) return -1;
    if (slot[(size_t)labels[i]] < 0) slot[(size_t)labels[i]] = (int32_t)i;
  }
  int64_t k = 0;
  for (int64_t l = 0; l < n_labels; ++l)
    if (slot[(size_t)l] >= 0) {
      first[k] = slot[(size_t)l];
      slot[(size_t)l] = (int32_t)k++;
    }
  for (int64_t i = 0; i < n; ++i) inv[i] = slot[(size_t)labels[i]];
  return k;
}

// Track / panel tables of the persistent window-chain kernels (GruProgram.chain_plan, temp_amd/gru_chain.py): every row of every
// instance of a chain gets a TRACK -- it inherits its predecessor's (prev >= 0), otherwise takes the lowest track no row of its
// instance inherits, new tracks when none is free -- and tracks are cut into panels of `T`; a panel's steps are the positions at
// which it has a row (panel-major, position-minor).
int temp_host_chain_tracks(int n_chains, const int64_t* chain_off, const int64_t* chain_inst, const int64_t* inst_n, const int64_t* inst_h0,
                           const int64_t* inst_rnn, const int64_t* prev_off, const int32_t* prev_cat, int T, int max_steps,
                           int64_t* counts, int32_t* panel, int32_t* rows, uint8_t* any_prev, int64_t* step_inst) {
  if (n_chains < 0 || T <= 0 || !counts || (n_chains > 0 && (!chain_off || !chain_inst || !inst_n || !inst_h0 || !inst_rnn || !prev_off)))
    return 2;
  const bool fill = panel != nullptr;
  if (fill && (!rows || !any_prev || !step_inst)) return 2;
  // The caller sizes its arrays from a first call without outputs, then calls again with them: that first call already builds
  // the tables into this thread's stash, and a second call with the SAME arguments only copies them out.
  struct Stash {
    bool valid = false;
    const void* key[8] = {};
    int n_chains = 0, T = 0, max_steps = 0;
    std::vector<int32_t> panel, rows;
    std::vector<uint8_t> any_prev;
    std::vector<int64_t> step_inst;
  };
  static thread_local Stash stash;
  const void* key[8] = {chain_off, chain_inst, inst_n, inst_h0, inst_rnn, prev_off, prev_cat, counts};
  if (fill && stash.valid && stash.n_chains == n_chains && stash.T == T && stash.max_steps == max_steps &&
      std::equal(key, key + 8, stash.key)) {
    stash.valid = false;                          // one use: the arrays behind the pointers may change after this
    std::copy(stash.panel.begin(), stash.panel.end(), panel);
    std::copy(stash.rows.begin(), stash.rows.end(), rows);
    std::copy(stash.any_prev.begin(), stash.any_prev.end(), any_prev);
    std::copy(stash.step_inst.begin(), stash.step_inst.end(), step_inst);
    counts[0] = (int64_t)(stash.panel.size() / 4); counts[1] = (int64_t)stash.step_inst.size();
    return 0;
  }
  stash.valid = false;
  const bool keep = !fill;                        // a sizing call: keep what the second call will ask for
  if (keep) { stash.panel.clear(); stash.rows.clear(); stash.any_prev.clear(); stash.step_inst.clear(); }
  int64_t P = 0, S = 0;
  std::vector<int64_t> prev_tr, tr, free_list;
  std::vector<uint8_t> used;
  std::vector<int32_t> tab;                       // [K][n_pan * T]
  for (int c = 0; c < n_chains; ++c) {
    const int64_t* ch = chain_inst + chain_off[c];
    const int64_t K = chain_off[c + 1] - chain_off[c];
    if (K <= 0) continue;
    const int64_t rnn = inst_rnn[ch[0]];
    for (int64_t k = 0; k < K; ++k)
      if (inst_rnn[ch[k]] != rnn) return 1;       // one GRU per chain
    // pass A: track of every row
    int64_t n_tracks = 0, total_rows = 0;
    for (int64_t k = 0; k < K; ++k) total_rows += inst_n[ch[k]];
    std::vector<int64_t> tracks((size_t)total_rows);
    std::vector<uint8_t> has_all((size_t)total_rows);
    int64_t base = 0;
    prev_tr.clear();
    for (int64_t k = 0; k < K; ++k) {
      const int64_t i = ch[k], n = inst_n[i];
      const int32_t* pi = (k > 0 && prev_cat) ? prev_cat + prev_off[i] : nullptr;
      tr.assign((size_t)n, -1);
      used.assign((size_t)n_tracks, 0);
      for (int64_t r = 0; r < n; ++r) {
        const bool has = pi && pi[r] >= 0;
        has_all[(size_t)(base + r)] = has ? 1 : 0;
        if (has) {
          if ((size_t)pi[r] >= prev_tr.size()) return 2;
          tr[(size_t)r] = prev_tr[(size_t)pi[r]];
          used[(size_t)tr[(size_t)r]] = 1;
        }
      }
      const int64_t nt0 = n_tracks;                // tracks that existed before this position: the free ones are reused first,
      int64_t f = 0;                               // lowest first; rows beyond them open new tracks
      for (int64_t r = 0; r < n; ++r) {
        if (tr[(size_t)r] >= 0) continue;
        while (f < nt0 && used[(size_t)f]) ++f;
        if (f < nt0) tr[(size_t)r] = f++;
        else tr[(size_t)r] = n_tracks++;
      }
      for (int64_t r = 0; r < n; ++r) tracks[(size_t)(base + r)] = tr[(size_t)r];
      prev_tr = tr;
      base += n;
    }
    if (n_tracks == 0) continue;
    const int64_t n_pan = (n_tracks + T - 1) / T;
    tab.assign((size_t)(K * n_pan * T), -1);
    base = 0;
    for (int64_t k = 0; k < K; ++k) {
      const int64_t i = ch[k], n = inst_n[i];
      for (int64_t r = 0; r < n; ++r)
        tab[(size_t)(k * n_pan * T + tracks[(size_t)(base + r)])] = (int32_t)((inst_h0[i] + r) | ((int64_t)has_all[(size_t)(base + r)] << 30));
      base += n;
    }
    for (int64_t p = 0; p < n_pan; ++p) {
      int64_t cnt = 0;
      const int64_t first = S;
      for (int64_t k = 0; k < K; ++k) {
        const int32_t* row = &tab[(size_t)(k * n_pan * T + p * T)];
        bool act = false, anyp = false;
        for (int t = 0; t < T; ++t)
          if (row[t] >= 0) { act = true; if ((row[t] >> 30) & 1) anyp = true; }
        if (!act) continue;
        if (fill) {
          for (int t = 0; t < T; ++t) rows[(size_t)S * T + t] = row[t];
          any_prev[S] = anyp ? 1 : 0;
          step_inst[S] = ch[k];
        } else if (keep) {
          stash.rows.insert(stash.rows.end(), row, row + T);
          stash.any_prev.push_back(anyp ? 1 : 0);
          stash.step_inst.push_back(ch[k]);
        }
        ++S; ++cnt;
      }
      if (cnt > max_steps) return 1;
      if (cnt > 0) {
        if (fill) { panel[4 * P] = (int32_t)rnn; panel[4 * P + 1] = (int32_t)first; panel[4 * P + 2] = (int32_t)cnt; panel[4 * P + 3] = 0; }
        else if (keep) { const int32_t e[4] = {(int32_t)rnn, (int32_t)first, (int32_t)cnt, 0}; stash.panel.insert(stash.panel.end(), e, e + 4); }
        ++P;
      }
    }
  }
  counts[0] = P; counts[1] = S;
  if (keep) {
    std::copy(key, key + 8, stash.key);
    stash.n_chains = n_chains; stash.T = T; stash.max_steps = max_steps;
    stash.valid = true;
  }
  return 0;
}

}  // extern "C"
