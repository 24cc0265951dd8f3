/* temp_amd host-side planner: plain C ABI, no GPU, no torch.
 *
 * The reference rebuilds its batched DGL graph (dgl.batch is C++ inside DGL) and its dense history tensors inside
 * every forward (models/DynamicRGCN.py:35-54,76-94,156-174).  Here the per-batch host work is index bookkeeping over
 * cached per-snapshot arrays; the two pieces that were interpreter-bound loops live in this library:
 *   temp_host_build_view   the sorted / chunked edge view of ONE snapshot (TempEdgeView layout of temp_amd.h);
 *   temp_host_chain_plan   the row maps of a window chain (which row of the previous executed position every node
 *                          continues from, and the time gap), i.e. get_prev_embeddings +
 *                          update_time_diff_hist_embeddings (models/DynamicRGCN.py:35-54) as int32 / float rows
 *                          instead of a dense re-zeroed (bsz, 2, N_ents, D) history.
 * All buffers are caller-owned; return value 0 = ok, non-zero = bad argument.
 */
#ifndef TEMP_AMD_HOST_H
#define TEMP_AMD_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Stable sort of the E edges by `seg` (0 <= seg < n_seg), every segment cut into chunks of <= `chunk` edges.
 *   order[E]                 the permutation (edge ids in view order)
 *   a_out[E], b_out[E]       a[order], b[order] as int32
 *   chunk_seg/beg/end/slot   one entry per chunk (capacity E): segment, edge range, partial-sum slot (-1 = single-chunk segment)
 *   fix_seg/slot/cnt         one entry per multi-chunk segment (capacity E): segment, first slot, number of chunks
 *   counts[3]                n_chunks, n_partial (chunks that go through slots), n_fix */
int temp_host_build_view(int64_t E, const int64_t* seg, const int64_t* a, const int64_t* b, int64_t n_seg, int64_t chunk,
                         int sort_b /* 1: a segment's edges in ascending b (ties: input order) -- the by-dst / by-src views, b = relation;
                                       0: input order -- the by-relation view */,
                         int64_t* order, int32_t* a_out, int32_t* b_out,
                         int32_t* chunk_seg, int32_t* chunk_beg, int32_t* chunk_end, int32_t* chunk_slot,
                         int32_t* fix_seg, int32_t* fix_slot, int32_t* fix_cnt, int64_t* counts);

/* Row maps of one chain (one direction) over its executed positions.
 *   n_steps executed positions; step s is window position pos[s] with n_win[s] active windows (windows 0 .. n_win[s]-1:
 *   left-padded windows form a suffix of the batch and, once active, stay active);
 *   gids[s * bsz + j] / gid_n[s * bsz + j]: global entity ids of window j's snapshot at step s (its node rows, in order).
 * Outputs, concatenated over steps in (step, window, node) order:
 *   prev_idx  row of the same entity in the PREVIOUS executed step's output of that window, -1 if it was not a node there
 *             (the reference's history is re-zeroed every position: only the immediately preceding step's nodes carry state)
 *   next_idx  the inverse map: which row of the NEXT executed step continues from this row, -1 if none (the last step's
 *             entries are all -1: its consumer, the target position, is planned by the caller)
 *   dt        pos[s] - (position at which the entity was last a node of the window, 0 if never)
 * and the final state after the last step: row_of[bsz * num_ents] (same meaning, for the target / all-entity consumers)
 * and last[bsz * num_ents] (last active position, 0 if never).  row_of / last are fully written. */
int temp_host_chain_plan(int bsz, int64_t num_ents, int n_steps, const int32_t* pos, const int32_t* n_win,
                         const int64_t* const* gids, const int64_t* gid_n,
                         int32_t* prev_idx, int32_t* next_idx, float* dt, int64_t* row_of, float* last);

/* Index lists of a batch's link-prediction loss (train_link_prediction + negative_sampling set-up,
 * models/TKG_Module.py:202-213, utils/CorrptTriples.py:36-56): for graph g with P_g chosen positives idx[g][0..P_g) the
 * 2 P_g rows [tail-corruption rows ; head-corruption rows] are appended to six int32 vectors of length R = 2 sum P_g
 * (packed[6][R]: known row, relation, is_tail, true entity (global id), lo, hi of the known-true slice) and weights[R] = 1/P_g;
 * triples[sum P_g][3] (local src, rel, dst) in graph order.
 *   graph_ptrs[g][8] = { src, rel, dst, gids (int64 arrays) , tail_lo, tail_hi, head_lo, head_hi (int32 arrays, per edge) }
 * pad4 != 0: every graph's block of 2 P_g rows is rounded up to a multiple of 4 rows with weight-0 rows (known = the graph's
 * first row, relation 0, empty known-true slice), so a block can be the N or K extent of an MFMA GEMM; R counts the padding. */
int temp_host_plan_loss(int n_graphs, const int64_t* graph_ptrs, const int64_t* const* idx, const int64_t* n_pos, const int64_t* row_offset,
                        int pad4, int64_t R, int32_t* packed, float* weights, int64_t* triples);

/* All three edge views of ONE snapshot (by destination, by source: chunk; by relation: chunk_rel) in the packed int32 layout
 * the device-side snapshot store keeps per snapshot:
 *   [by_dst: a b chunk_seg chunk_beg chunk_end chunk_slot fix_seg fix_slot fix_cnt][by_src: ...][by_rel: ...]
 *   [rel_rank: rank of every by-relation chunk inside its relation][in_deg n][out_deg n][nnorm bits n]
 * by_dst: seg = dst, a = src, b = rel;  by_src: seg = src, a = dst, b = rel;  by_rel: seg = rel, a = src, b = dst.
 * sizes[31] = length of each array in that order, n_partial[3], rel_chunks[n_rel_rows] = chunks per relation.
 * packed must hold 28 * max(E, 1) + 3 * n entries.  Returns the number of entries written, or -1. */
int64_t temp_host_snapshot_pack(int64_t n, int64_t E, const int64_t* src, const int64_t* dst, const int64_t* rel, const float* nnorm,
                                int64_t n_rel_rows, int64_t chunk, int64_t chunk_rel,
                                int32_t* packed, int64_t* sizes, int64_t* n_partial, int64_t* rel_chunks);

/* Control block of temp_assemble_views (include/temp_amd.h) for a batch = disjoint union of M member snapshots whose packed
 * views are resident on the device.  meta[m] (int64, 66 + n_rel_rows entries) = { size[31], offset[31] of the member's arrays
 * inside its packed buffer, device address of that buffer, n_partial[3], chunks per relation[n_rel_rows] }; node_off / edge_off
 * [M] = first node / edge of every member inside the union.
 * ctl (int32) receives  [descriptors (8 words each) | piece_desc | piece_start | slot table M x n_rel_rows | fix_seg | fix_slot | fix_cnt]
 * and summary (int64[72]) = { [0] n_desc, [1] n_pieces, [2] n_fix, [3] tail0 = words of the member-fed arrays,
 * [4..31] first word of each of the 27 member-fed output arrays + their end, [35..61] their lengths,
 * [65..67] partial slots of by_dst, by_src, by_rel }.  The 27 arrays, in output order:
 * in_deg, out_deg, nnorm, the nine by_dst arrays, the nine by_src arrays, by_rel a / b / chunk_seg / chunk_beg / chunk_end /
 * chunk_slot.  Returns the number of ctl words written, or -1 (bad argument / ctl_cap too small). */
int64_t temp_host_union_plan(int64_t M, int64_t n_rel_rows, const int64_t* meta, const int64_t* node_off, const int64_t* edge_off,
                             int64_t piece, int32_t* ctl, int64_t ctl_cap, int64_t* summary);

/* k distinct integers of [0, n), uniformly at random, in random order (np.random.choice(n, k, replace=False) /
 * torch.randperm(n)[:k] of the reference: models/DynamicRGCN.py:81, utils/CorrptTriples.py:38-40): a partial Fisher-Yates
 * shuffle driven by splitmix64(seed).  out[k]. */
int temp_host_sample_subset(int64_t n, int64_t k, uint64_t seed, int64_t* out);

/* Inverse of a gather index list (the static maps the deterministic segment-sum adjoints reduce over, temp_segment_sum_rows):
 * positions i of idx[0..n) grouped by table row idx[i] (stable, entries < 0 skipped):
 *   seg_ptr[n_rows + 1], order[count of non-negative entries];  returns that count, or -1 on a bad argument. */
int64_t temp_host_gather_inverse(int64_t n, const int64_t* idx, int64_t n_rows, int32_t* seg_ptr, int32_t* order);

/* Distinct labels of labels[0..n) (all in [0, n_labels)) in ascending label order -- numpy.unique(labels, return_index=True,
 * return_inverse=True) in O(n + n_labels): first[k] = position of the first occurrence of the k-th distinct label, inv[i] = k
 * of labels[i].  (Which chain rows share their GRU input gates: rows gathered from the same layer-output row,
 * temp_amd/gru_chain.py GruProgram.gi_shared.)  first[n], inv[n]; returns the number of distinct labels, or -1 on a bad argument. */
int64_t temp_host_unique_labels(int64_t n, const int64_t* labels, int64_t n_labels, int32_t* first, int32_t* inv);

/* Track / panel tables of the persistent window-chain kernels (include/temp_amd.h: TempGruChain; replaces the per-position
 * history bookkeeping of models/DynamicRGCN.py:35-54 for the chain kernels).  Chain c = instances chain_inst[chain_off[c] ..
 * chain_off[c+1]) in position order; instance i has inst_n[i] rows starting at row inst_h0[i] of the step's row space and uses GRU
 * inst_rnn[i]; prev_cat[prev_off[i] + r] = row (inside the previous instance of the chain) that row r continues from, or -1
 * (ignored for the first instance of a chain).  A row inherits the track of its predecessor, otherwise takes the lowest track
 * no row of its instance inherits; tracks are cut into panels of T.
 *   pass 1 (panel == NULL): counts[2] = { panels P, steps S };  pass 2: panel[P][4] = { rnn, first step, steps, 0 },
 *   rows[S][T] = row | (has_prev << 30) or -1, any_prev[S], step_inst[S] = instance of the step (panel-major, position-minor).
 *   Pass 1 keeps its tables in thread-local storage; a pass 2 from the same thread with the very same argument pointers (the
 *   usual size-then-fill sequence, inputs untouched in between) copies them out instead of planning again.
 * Returns 0, 1 = not representable (a chain mixes GRUs, or a panel has more than max_steps steps), 2 = bad argument. */
int temp_host_chain_tracks(int n_chains, const int64_t* chain_off, const int64_t* chain_inst, const int64_t* inst_n, const int64_t* inst_h0,
                           const int64_t* inst_rnn, const int64_t* prev_off, const int32_t* prev_cat, int T, int max_steps,
                           int64_t* counts, int32_t* panel, int32_t* rows, uint8_t* any_prev, int64_t* step_inst);

int temp_host_abi_version(void);
#ifdef __cplusplus
}
#endif
#endif
