// Host-side planner (see include/temp_amd_host.h).  Plain C++17; built into temp_amd/libtemp_host.so by temp_amd/build.py.
#include "temp_amd_host.h"
#include <cstddef>
#include <vector>

extern "C" {

int temp_host_abi_version(void) { return 1; }

int temp_host_build_view(int64_t E, const int64_t* seg, const int64_t* a, const int64_t* b, int64_t n_seg, int64_t chunk,
                         int64_t* order, int32_t* a_out, int32_t* b_out,
                         int32_t* chunk_seg, int32_t* chunk_beg, int32_t* chunk_end, int32_t* chunk_slot,
                         int32_t* fix_seg, int32_t* fix_slot, int32_t* fix_cnt, int64_t* counts) {
  if (E < 0 || n_seg < 0 || chunk <= 0 || !counts || (E > 0 && (!seg || !a || !b || !order || !a_out || !b_out))) return 1;
  std::vector<int64_t> ptr((size_t)n_seg + 1, 0);
  for (int64_t e = 0; e < E; ++e) {
    if (seg[e] < 0 || seg[e] >= n_seg) return 2;
    ++ptr[(size_t)seg[e] + 1];
  }
  for (int64_t s = 0; s < n_seg; ++s) ptr[(size_t)s + 1] += ptr[(size_t)s];
  {                                                    // counting sort: stable, edges keep their order inside a segment
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) order[cur[(size_t)seg[e]]++] = e;
  }
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
                         int64_t* prev_idx, float* dt, int64_t* row_of, float* last) {
  if (bsz < 0 || num_ents < 0 || n_steps < 0 || !row_of || !last || (n_steps > 0 && (!pos || !n_win || !gids || !gid_n))) return 1;
  const size_t total = (size_t)bsz * (size_t)num_ents;
  for (size_t i = 0; i < total; ++i) { row_of[i] = -1; last[i] = 0.f; }
  int64_t out = 0;
  int prev_step = -1;
  for (int s = 0; s < n_steps; ++s) {
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
        prev_idx[out + i] = ro[e];
        dt[out + i] = p - la[e];
      }
      out += n;
    }
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
                        int64_t R, int32_t* packed, float* weights, int64_t* triples) {
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
    if (P < 0 || row + 2 * P > R) return 2;
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
    trow += P;
  }
  return row == R ? 0 : 3;
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

}  // extern "C"
