#!/usr/bin/env python3
"""Headline benchmark: edges/sec of a train-seq-len=15 RGCN+GRU window encoder, forward+backward.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S-gdelt]

One "step" = one forward+backward pass of the snapshot encoder (2 RGCN layers + GRU/BiGRU chain,
BiGRRGCN with --rec-only-last-layer) over one batch of bsz windows of synthetic GDELT-shaped
snapshots (temp_amd/synthetic.py; the real GDELT files are not shipped with the reference), with
the upstream gradient = ones on the target-position output.  A unit of work is one
snapshot-edge visit; `value` = edge visits processed by all ranks / wall time, inputs (parameters,
snapshot edge views, window row maps) already resident in HBM.

Multi-GPU (one process per GPU, torch.distributed over RCCL): every rank encodes its own bsz
windows (weak scaling, the reference's DDP axis) and the parameter gradients are all-reduced in
one bucket after backward.

Extra objects on the JSON line:
  extra         short secondary measurements of the same process (never the headline): the step with the link-prediction loss,
                the self-attention encoder (config 5), the snapshot-sharded north-star step on one RCCL rank, BASELINE configs 1-3
                at their own sizes (static RGCN; GRRGCN with the reference's default flags; post-ensemble BiGRRGCN), and the FULL
                window in the HBM regime (S-hbm-window: bi L=15 bsz=1 over 2^18-node / 2^22-edge snapshots, 230 relations --
                the regime BASELINE's "% HBM roofline" is named after; its own roofline object uses SURVEY 8d's byte model)
  roofline      dominant kernel (largest share of traced kernel time): algorithmic bytes (or
                flops) per step / its HIP-event time per step, against 8 TB/s HBM or the MFMA peak
                of the pipe the kernel runs on: 157.3 TFLOP/s fp32 MFMA, or -- for the large GEMMs,
                which compute every fp32 product as six bf16 MFMA products of an exact three-way
                operand split -- 2500 / 6 TFLOP/s (roofline.pipe says which; `achieved` always
                counts the ALGORITHMIC fp32 flops).  config.fp32_mfma_ms_per_step is the same step
                with every product on the fp32 MFMA kernels (temp_set_option(TEMP_OPT_MFMA_BF16X3, 0), same
                process, re-captured graph).  Events are recorded by libtemp_amd around every launch
                on the launch stream (temp_trace_begin/end) during extra traced steps run right
                after the timed region, so the headline number is not perturbed.
  cpu_baseline  the CPU oracle (torch restatement of the reference's op sequence, kind "port")
                timed on the host cores on ONE window of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak, same guide
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak, same guide
# The large GEMMs (k_gemm_panel<*> = k_gemm_bxr / k_gemm_bxp / k_gemm_bx, k_gemm_tn = k_gemm_tn_bx8 / k_gemm_tn_bx) compute every fp32 product as SIX bf16 MFMA
# products of an exact three-way operand split (temp_amd/csrc/gemm_bx.hpp; fp32-equivalent accuracy): their roof is the bf16 pipe
# divided by six.  temp_set_option(TEMP_OPT_MFMA_BF16X3, 0) (or TEMP_MFMA=f32 in the environment at load time) keeps them on the
# fp32 MFMA kernels (round-1 arithmetic); MFMA_MODE is read from the library in main().
BX_KERNELS = ("k_gemm_panel", "k_gemm_tn_bx", "k_gru_chain_fwd", "k_gru_chain_bwd", "k_gru_wgrad")   # (k_gemm_tn itself is the fp32 MFMA kernel of the small products; the chain kernels run the split products when d % 8 == 0)
# Round 6: where an f16 kernel exists (temp_amd/csrc/split_f16.hpp) the same products run as THREE f16 MFMA products of the scaled
# two-way split -- the window-chain kernels, the GRU weight gradients, the input-gate and d_x GEMMs (their operands arrive with
# magnitude keys from their producers); the roof of those kernels is the 16-bit pipe divided by three.  The self-loop products and
# the loop-weight gradient (no keys yet) stay on the six-product bf16 kernels.
HX_KERNELS = ("k_gru_chain_fwd", "k_gru_chain_bwd", "k_gru_wgrad", "k_gemm_panel<gru_gi>", "k_gemm_panel<gru_dx>")
MFMA_MODE = "bf16x3"           # "f32" | "bf16x3" | "f16x2" (main() reads the library's options)
OPT_MFMA_BF16X3 = 0            # include/temp_amd.h: TEMP_OPT_MFMA_BF16X3
OPT_MFMA_F16X2 = 9             # include/temp_amd.h: TEMP_OPT_MFMA_F16X2
CPU_THREADS = 16               # cpu_baseline leg (--cpu-threads)


def profile_file(stem):
    """profiles/<round>_<stem> of the latest round that has it (static counter summaries: rocprofv3 --pmc runs cannot be taken from
    inside this process)."""
    for tag in ("r06", "r05", "r04"):
        p = os.path.join(REPO, "profiles", "%s_%s" % (tag, stem))
        if os.path.exists(p):
            return p
    return os.path.join(REPO, "profiles", "r06_%s" % stem)



def make_args(w, module, dropout=0.0):
    dropout = float(os.environ.get("TEMP_BENCH_DROPOUT", dropout))       # (experiments: any configuration with the self-loop dropout drawing; run with --no-graph)
    return argparse.Namespace(
        n_bases=w["B"], dropout=dropout, inv_temperature=0.1, learnable_lambda=False, impute=False, post_aggregation=False,
        post_ensemble=False, num_layers=1, type1=False, rec_only_last_layer=True, use_time_embedding=False, module=module,
        embed_size=w["D"], hidden_size=w["D"], num_pos_facts=3000, negative_rate=500, score_function="complex",
        train_seq_len=w["L"], test_seq_len=w["L"], use_cuda=True, debug=True, edge_dropout=False, random_dropout=False,
        use_embed_for_non_active=False, lr=1e-3, seed=0, batch_size=w["bsz"])


def build_model(w, device, encoder="gru", dropout=0.0):
    from temp_amd.bi_dynamic_rgcn import BiDynamicRGCN
    from temp_amd.dynamic_rgcn import DynamicRGCN
    from temp_amd.self_attention_rgcn import BiSelfAttentionRGCN, SelfAttentionRGCN
    torch.manual_seed(1)
    bi = w["module"].startswith("Bi")
    if encoder == "attention":
        cls = BiSelfAttentionRGCN if bi else SelfAttentionRGCN
    else:
        cls = BiDynamicRGCN if bi else DynamicRGCN
    snaps = w["snapshots"]
    m = cls(make_args(w, w["module"], dropout), w["num_ents"], w["num_rels"], snaps, snaps, snaps)
    return m.to(device)


def train_loop(model, w, steps):
    """Wall time of a training loop in which EVERY step is a new window batch (what main.py's trainer does): host-side
    prepare, fresh negative samples (one kernel launch), link-prediction loss, backward, Adam; eager launches.  Timed twice:
    batches prepared by the background prefetcher (temp_amd.prefetch.BatchPrefetcher, the loop the package documents) and
    prepared inline.  Reported next to the headline (which replays one resident batch as a HIP graph), never as the headline."""
    from temp_amd import synthetic
    from temp_amd.prefetch import BatchPrefetcher
    from temp_amd.sampling import CorruptTriples
    if not hasattr(model, "corrupter"):
        model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
    WARM = 10                               # untimed steps at the head of each loop (worker threads, their streams and pinned rings start here)
    opt = model.configure_optimizers()      # the reference trainer's: Adam(lr, weight_decay=1e-4), models/TKG_Module.py:154-160
    batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 1000 + r) for r in range(steps + WARM)]
    for b in batches:                       # the first visit of a snapshot builds and uploads its cached views: once per run
        model.prepare(b, w["L"], True)

    def timed(source):
        edges, t0 = 0, None
        for i, wb in enumerate(source):
            if i == WARM:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            loss = model.run_loss(wb)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            if i >= WARM:
                edges += wb.n_edge_visits
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return 1e3 * dt / steps, edges / dt

    inline_ms, inline_eps = timed(model.prepare(b, w["L"], True) for b in batches)
    pre_ms, pre_eps = timed(BatchPrefetcher(model, batches, seq_len=w["L"], depth=4, workers=2))
    best = "prefetcher" if pre_ms <= inline_ms else "inline"     # which one wins depends on the load of the box's host CPUs (one GIL)
    return dict(ms_per_step=min(pre_ms, inline_ms), edges_per_s=max(pre_eps, inline_eps), mode=best, prefetcher_ms_per_step=pre_ms,
                prefetcher_edges_per_s=pre_eps, inline_prepare_ms_per_step=inline_ms, inline_prepare_edges_per_s=inline_eps, steps=steps,
                what="new batch every step: host prepare (the faster of: background prefetcher with 2 workers / prepared inline, both reported) + negatives + loss + backward "
                     "+ Adam, eager launches (host-bound)")


def algorithmic_costs(wb, D, bi, S=2):
    """Per-step ALGORITHMIC bytes / flops of every kernel family of the batched path (fp32, int32
    ids).  n = node visits, E = edge visits of the step; the GRU runs once per node visit (twice on
    the target position of the bi model)."""
    if not hasattr(wb, "target"):                 # attention encoder: RGCN rows only (the mixer is reported by the kernel table)
        bi = False
    n_gru = wb.n_node_visits + (wb.target.n_rows if bi else 0)     # GRU cells: one per node visit (two on the bi target)
    # the RGCN layers run once per DISTINCT snapshot of the step (shared between overlapping windows)
    n = getattr(wb, "n_nodes_distinct", 0) or wb.n_node_visits
    E = getattr(wb, "n_edges_distinct", 0) or wb.n_edge_visits
    row = 4 * D
    c = {}
    nv = wb.n_node_visits
    c["k_gather_rows"] = dict(bytes=(n + nv) * (4 + 2 * row), flops=0)
    c["k_scatter_add_rows"] = dict(bytes=(n + nv) * (4 + 3 * row), flops=0)
    c["k_segment_sum_rows"] = dict(bytes=(n + nv) * (4 + 2 * row), flops=0)      # the deterministic gather adjoints (traced as k_scatter_add_rows until round 6)
    c["k_absmax_keys"] = dict(bytes=0, flops=0)
    c["k_rgcn_agg<fwd>"] = dict(bytes=2 * (E * (row + 8) + n * row), flops=2 * E * 2 * D * S)
    c["k_rgcn_agg<dx>"] = dict(bytes=2 * (E * (row + 12) + n * row), flops=2 * E * 2 * D * S)
    c["k_rgcn_dw"] = dict(bytes=2 * (E * (2 * row + 12)), flops=2 * E * 2 * D * S)
    # self-loop GEMMs: layer 2 over the n node rows; layer 1 is fed by the embedding gather, so its self-loop product and
    # both of its gradients run over the N_ents table rows (temp_rgcn_table_fwd/bwd)
    nt = getattr(wb, "n_table_rows", 0)
    c["k_gemm_panel<loop_fwd>"] = dict(bytes=n * 3 * row, flops=2 * n * D * D)
    c["k_gemm_panel<loop_dx>"] = dict(bytes=n * 3 * row + nt * 3 * row, flops=2 * (n + nt) * D * D)
    c["k_gemm_panel<isolated>"] = dict(bytes=nt * 3 * row, flops=2 * nt * D * D)
    # weight gradients, one entry per KERNEL (trace ids follow the kernels since round 4): the four 3d x d products of the two GRUs
    # (k_gemm_tn_bx8: d_W_ih = dgi^T x, d_W_hh = dgh^T hdec), layer 2's loop weight (k_gemm_tn_bx), the table layer's (k_gemm_tn)
    c["k_gemm_tn_bx8"] = dict(bytes=n_gru * (2 * row + 6 * row), flops=2 * 2 * n_gru * 3 * D * D)
    # round 5: the same four products from the ONE gate-gradient matrix g4 = [dr | dz | dn_i | dn_h] (k_gru_wgrad): x, hdec and the
    # 4 d gate columns once each (the shared [dr dz] columns are counted once: what an ideal kernel reads)
    c["k_gru_wgrad"] = dict(bytes=n_gru * (2 * row + 4 * row), flops=2 * 2 * n_gru * 3 * D * D)
    c["k_gemm_tn_bx"] = dict(bytes=n * 2 * row, flops=2 * n * D * D)
    c["k_gemm_tn"] = dict(bytes=nt * 2 * row, flops=2 * nt * D * D)
    c["k_relu_bwd"] = dict(bytes=n * 3 * row, flops=0)
    c["k_gru_fwd"] = dict(bytes=n_gru * (row + 3 * row + row + 5 * row + 8), flops=6 * n_gru * D * D)   # hoisted: h-phase only
    c["k_gemm_panel<gru_gi>"] = dict(bytes=n_gru * (row + 3 * row), flops=2 * n_gru * 3 * D * D)
    c["k_gru_bwd_gates"] = dict(bytes=n_gru * (6 * row + 6 * row), flops=0)
    c["k_gemm_panel<gru_dx>"] = dict(bytes=n_gru * (3 * row + row), flops=2 * n_gru * 3 * D * D)
    c["k_gemm_panel<gru_dprev>"] = dict(bytes=n_gru * (3 * row + 3 * row), flops=2 * n_gru * 3 * D * D)
    c["k_colsum_part"] = dict(bytes=n_gru * 6 * row, flops=0)
    # persistent window chain (one launch per direction of time for ALL positions): forward reads gi (3 rows) and writes h + 5 saved
    # planes; backward reads the 5 planes (+ upstream rows) and writes dgi + dgh (6 rows); 6 n D^2 flop each (W_hh product)
    c["k_gru_chain_fwd"] = dict(bytes=n_gru * (3 * row + 6 * row), flops=6 * n_gru * D * D)
    c["k_gru_chain_bwd"] = dict(bytes=n_gru * (5 * row + 4 * row), flops=6 * n_gru * D * D)      # (round 5: g4 = 4 gate rows, was dgi + dgh = 6)
    c["k_gru_chain_pack"] = dict(bytes=4 * 2 * 3 * D * D * 4, flops=0)
    c["k_bx_pack"] = dict(bytes=(2 * D * D + 2 * 2 * 3 * D * D) * (4 + 6), flops=0)      # weights in as fp32, out as three bf16 planes
    if hasattr(wb, "idx_tgt"):                    # attention mixer over the target rows (encoder-only step)
        nq, act = int(wb.idx_tgt.shape[0]), int((wb.idx_tgt >= 0).sum().item())
        c["k_sa_attn_fwd"] = dict(bytes=nq * 4 * row + act * 2 * row, flops=4 * (act + nq) * D)
        c["k_sa_attn_bwd"] = dict(bytes=nq * 8 * row + act * 6 * row, flops=8 * (act + nq) * D)
        R = wb.n_hist_rows
        c["k_gemm_panel<linear>"] = dict(bytes=(R * 3 + nq * 4) * row * 2, flops=2 * (2 * R * 2 + 2 * nq * 3) * D * D)
    return c


MFMA_KERNELS = ("k_gemm_panel", "k_gemm_tn", "k_gru_fwd", "k_gru_chain_fwd", "k_gru_chain_bwd", "k_gru_wgrad")   # (prefixes)


def traced_steps(step_fn, n_steps, lib):
    """Per-kernel HIP-event times of n_steps eager steps.  One trace per step; ONE EXTRA first step is run and dropped (on a fresh
    box the first eager launch of a kernel pays its code-object load: round 5's driver line showed k_gru_wgrad at 2.68 ms in a
    3-step mean against 0.218 ms), and every kernel's per-step time is the MEDIAN over the kept steps."""
    cap = 20000
    from temp_amd import _lib
    per_step = []
    for it in range(n_steps + 1):
        _lib.check(lib.temp_trace_begin(cap), "temp_trace_begin")
        step_fn()
        ids = (ctypes.c_int32 * cap)()
        ms = (ctypes.c_float * cap)()
        n = ctypes.c_int32(0)
        _lib.check(lib.temp_trace_end(ids, ms, cap, ctypes.byref(n)), "temp_trace_end")
        if it == 0:
            continue
        agg = {}
        for i in range(n.value):
            name = lib.temp_trace_kernel_name(ids[i]).decode()
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += ms[i]
        per_step.append(agg)
    out = {}
    for k in sorted({k for agg in per_step for k in agg}):
        launches = float(np.median([agg.get(k, [0, 0.0])[0] for agg in per_step]))
        tot = float(np.median([agg.get(k, [0, 0.0])[1] for agg in per_step]))
        if launches > 0:
            out[k] = dict(launches_per_step=launches, ms_per_step=tot, avg_ms=tot / launches)
    return out


def _oracle_batch_runner(om, cfg, w, gd, seed=7):
    """-> run(t_list): one oracle step (forward + backward, upstream gradient = ones) over ONE batch of windows with the
    target graphs cut to a random 50 % of their edges and re-normalised, as the training step does (models/DynamicRGCN.py:76-90)
    -> (edge visits, snapshot visits)."""
    from oracle import temp_oracle as O
    bi = w["module"].startswith("Bi")
    times = sorted(gd.keys())
    L = w["L"]
    leaves = list(O.leaf_tensors(om).values())
    for v in leaves:
        v.requires_grad_(True)
    rng = np.random.default_rng(seed)

    def half(g):
        E = g.num_edges
        return O.edge_subgraph(g, np.sort(rng.choice(E, size=E // 2, replace=False))) if E > 1 else g

    def run(t_list):
        for v in leaves:
            v.grad = None
        tgt = [half(gd[t]) for t in t_list]
        if bi:
            tf, tb = O.get_batch_graph_list_bi(list(t_list), L, times)
            Hf = O.bi_pre_forward(om, cfg, gd, tf, L, True)
            Hb = O.bi_pre_forward(om, cfg, gd, tb, L, False)
            out = O.bi_target_embeds(om, cfg, Hf, Hb, tgt, tf[-1], L)
            hist = [x for col in tf[:-1] + tb[:-1] for x in col if x is not None]
        else:
            tf = O.get_batch_graph_list(list(t_list), L, times)
            H = O.uni_pre_forward(om, cfg, gd, tf, L)
            out = O.uni_target_embeds(om, cfg, H, tgt, tf[-1], L)
            hist = [x for col in tf[:-1] for x in col if x is not None]
        sum(o.sum() for o in out).backward()
        return sum(gd[x].num_edges for x in hist) + sum(g.num_edges for g in tgt), len(hist) + len(tgt)

    return run


def cpu_baseline(model, w, device_targets, min_seconds=12.0, max_batches=16):
    """The oracle (port of the reference's op sequence, dense re-zeroed history kept) on the host cores, on the SAME batch the
    GPU step encodes: the rank-0 targets as ONE batch of bsz windows, target graphs at 50 % of their edges, fwd+bwd, repeated
    until at least `min_seconds` of CPU work has been timed.  `all_cores` is a second, bounded measurement with every hardware
    CPU this process may use (_cpu_quota: the cgroup's quota, 16 on the GPU boxes whose hosts show 256 threads): several processes of
    16 torch threads side by side when the quota allows it (cpu_all_cores_parallel), else the figure above IS all of them."""
    from oracle import temp_oracle as O
    nthreads = min(os.cpu_count() or 1, CPU_THREADS)
    torch.set_num_threads(nthreads)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cfg = dict(module=w["module"], n_bases=w["B"], inv_temperature=0.1, rec_only_last_layer=True, use_time_embedding=False)
    om = O.model_from_state_dict(sd, cfg)
    gd = {t: O.SnapGraph(g.n, g.src, g.dst, g.rel, g.gids) for t, g in w["snapshots"].items()}
    run = _oracle_batch_runner(om, cfg, w, gd)
    run(device_targets[:1])                         # warm-up (allocator, thread pool)
    edges = visits = nb = 0
    t0 = time.perf_counter()
    while True:
        e, v = run(device_targets)
        edges, visits, nb = edges + e, visits + v, nb + 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or nb >= max_batches:
            break
    quota = _cpu_quota()
    if quota >= 2 * nthreads:                       # more CPUs than one process uses well: several processes side by side
        allc = cpu_all_cores_parallel(w, threads_per_process=nthreads, processes=quota // nthreads)
    else:
        allc = dict(cores=quota, value=edges / dt, unit="edges/s", same_as="value",
                    note="this process may use %d CPUs at once (cgroup CPU quota; the host shows %d hardware threads): the %d-thread figure is every "
                         "core it has.  More busy threads are throttled -- measured with tools/cpu_scale_probe.py: 16 processes x 16 threads deliver "
                         "0.07 M edge visits/s, ONE process with 256 torch threads does not finish a window in 45 s" % (quota, os.cpu_count() or 1, nthreads))
    return dict(value=edges / dt, unit="edges/s", cores=nthreads, host_threads=os.cpu_count(), cpu_quota_cores=quota, cpu_model=_cpu_model(), kind="port",
                all_cores=allc,
                all_cores_one_process=cpu_all_cores_probe(w) if os.environ.get("TEMP_BENCH_CPU_ONE_PROCESS_PROBE") else None,
                sample="%d batches of %d windows of %s (the step's own targets, target graphs at 50 %% of their edges): %d snapshot visits, "
                       "%d edge visits, fwd+bwd, %.1f s" % (nb, len(device_targets), w["name"], visits, edges, dt))


def cpu_probe_main(a):
    """Child process of cpu_all_cores_probe: ONE window (bsz = 1) of the workload through the oracle with a.cpu_threads torch
    threads; prints one JSON line.  Needs no GPU (oracle-initialised parameters of the same shapes: timing only)."""
    from oracle import temp_oracle as O
    from temp_amd import synthetic
    torch.set_num_threads(max(1, a.cpu_threads))
    w = synthetic.workload(a.workload, seed=0)
    cfg = dict(module=w["module"], n_bases=w["B"], inv_temperature=0.1, rec_only_last_layer=True, use_time_embedding=False)
    om = O.init_model(cfg, w["num_ents"], w["num_rels"], w["num_times"], w["D"], seed=1)
    gd = {t: O.SnapGraph(g.n, g.src, g.dst, g.rel, g.gids) for t, g in w["snapshots"].items()}
    run = _oracle_batch_runner(om, cfg, w, gd, seed=7 + a.cpu_probe_rank)
    if a.cpu_probe_seconds > 0:
        # one of several processes that share the host (cpu_all_cores_parallel): whole batches of bsz windows, this process's own
        # targets, until the time is up; every finished batch is reported at once (a process stopped early has still counted)
        t = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], a.cpu_probe_rank)
        run(t[:1])                                  # warm-up
        print(json.dumps(dict(stage="ready")), flush=True)
        t0 = time.perf_counter()
        while True:
            tb = time.perf_counter()
            e, v = run(t)
            now = time.perf_counter()
            print(json.dumps(dict(stage="batch", seconds=now - tb, edges=e, visits=v, t_end=now - t0)), flush=True)
            if now - t0 >= a.cpu_probe_seconds:
                break
        return
    t = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)[:1]
    print(json.dumps(dict(stage="ready", edges_per_window=None)), flush=True)
    t0 = time.perf_counter()
    e, v = run(t)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(stage="done", seconds=dt, edges=e, visits=v, threads=torch.get_num_threads())), flush=True)


def cpu_all_cores_parallel(w, threads_per_process=16, seconds=14.0, timeout_s=75.0, processes=None):
    """Every CPU this process may use on the oracle, the way it scales: `processes` (default cpu_count / 16) PROCESSES of 16 torch
    threads each (the measured optimum of one process), each encoding whole batches of bsz windows (its own targets, fwd + bwd) for `seconds`;
    value = all edge visits of the batches that finished / the span they finished in; per-batch times give the median.  The
    intra-op pool of ONE process does not use 256 threads (cpu_all_cores_probe: one window does not finish in 45 s)."""
    import subprocess
    n = os.cpu_count() or 1
    procs = max(1, processes if processes else n // threads_per_process)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads_per_process), MKL_NUM_THREADS=str(threads_per_process))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-probe", "--cpu-threads", str(threads_per_process), "--workload", w["name"],
           "--cpu-probe-seconds", str(seconds)]
    t0 = time.perf_counter()
    ps = []
    try:
        for r in range(procs):
            ps.append(subprocess.Popen(cmd + ["--cpu-probe-rank", str(r)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True))
    except OSError as e:
        for p in ps:
            p.kill()
        return dict(cores=n, value=None, note="probe not started: %s" % e)
    outs = []
    for p in ps:
        left = max(1.0, timeout_s - (time.perf_counter() - t0))
        try:
            out, _ = p.communicate(timeout=left)
        except subprocess.TimeoutExpired:
            p.kill()                                # exactly the children started above
            out, _ = p.communicate()
        outs.append(out or "")
    batches, edges, span = [], 0, 0.0
    for out in outs:
        for line in out.splitlines():
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if d.get("stage") == "batch":
                batches.append(d["seconds"])
                edges += d["edges"]
                span = max(span, d["t_end"])
    if not batches:
        return dict(cores=n, processes=procs, value=None, unit="edges/s", note="no batch finished within %.0f s" % timeout_s)
    med = float(np.median(batches))
    return dict(cores=procs * threads_per_process, processes=procs, threads_per_process=threads_per_process, value=edges / span, unit="edges/s",
                batches=len(batches), median_batch_s=med, span_s=span,
                sample="%d processes x %d torch threads, each whole batches of %d windows of %s (its own targets, target graphs at 50 %% of their edges), "
                       "fwd+bwd: %d batches, %d edge visits in %.1f s; median batch %.2f s" % (procs, threads_per_process, w["bsz"], w["name"], len(batches), edges, span, med))


def cpu_all_cores_probe(w, timeout_s=45.0):
    """The oracle with torch.set_num_threads(os.cpu_count()) on ONE window, measured NOW in a child process that is stopped
    after `timeout_s` (round 2 measured 598 s per window with 256 threads: the probe then reports the bound it established)."""
    import subprocess
    n = os.cpu_count() or 1
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-probe", "--cpu-threads", str(n), "--workload", w["name"]]
    env = dict(os.environ, OMP_NUM_THREADS=str(n), MKL_NUM_THREADS=str(n))
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
    except OSError as e:
        return dict(cores=n, value=None, note="probe not started: %s" % e)
    ready_at, done = None, None
    try:
        out, _ = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.kill()                                    # exactly the child started above
        out, _ = p.communicate()
    for line in (out or "").splitlines():
        try:
            d = json.loads(line)
        except ValueError:
            continue
        if d.get("stage") == "done":
            done = d
    wall = time.perf_counter() - t0
    N, R, E, nn, T, D, B, L, bsz, module = __import__("temp_amd.synthetic", fromlist=["WORKLOADS"]).WORKLOADS[w["name"]]
    per_window = E * ((2 * L - 1) if module.startswith("Bi") else L)
    if done is not None:
        return dict(cores=n, value=done["edges"] / done["seconds"], unit="edges/s", seconds=done["seconds"], sample="1 window (bsz 1), full target graph halved, fwd+bwd")
    return dict(cores=n, value=None, unit="edges/s", timed_out_after_s=round(wall, 1), value_upper_bound=per_window / max(wall, 1e-9),
                note="ONE window did not finish within the time limit with every hardware thread (torch's intra-op pool turns the oracle's "
                     "many small ops into barrier traffic); value_upper_bound = edge visits of a full window / the time allowed, incl. start-up")


def measure_sharded(model, w, world, rank, device, steps, warmup, dist, graphs=True):
    """BASELINE north_star variant (config 4): ONE global batch of bsz * world windows; its distinct snapshots are cut into
    `world` edge-balanced shards, each rank runs the two RGCN layers on its shard, one direct all-gather over xGMI hands every
    rank all per-snapshot node states, the recurrent chain is sharded by window, gradients are all-reduced in one bucket.
    Everything except the collectives is replayed from three HIP graphs (temp_amd.dist.ShardedStep).
    -> dict for the JSON line (rank 0) or None."""
    from temp_amd import synthetic
    from temp_amd.dist import ShardedStep, SnapshotShardedEncoder, allgather_rows
    all_targets = [t for r in range(world) for t in synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r)]
    model.sample_rng = np.random.default_rng(2)
    enc = SnapshotShardedEncoder(model)
    sb = enc.prepare(all_targets, w["L"], train=True)
    params = [p for p in model.parameters()]
    st = ShardedStep(enc, sb, params, graphs=graphs, average=False, force_allreduce=True)    # (force: a single forced rank still calls RCCL)

    for _ in range(warmup):
        st.step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step()
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # the all-gather alone (forward direction) on this step's buffers
    n_local = sb.row_bounds[rank + 1] - sb.row_bounds[rank]
    y = torch.randn(n_local, w["D"], device=device)
    for _ in range(3):
        allgather_rows(y, st.y2_all, sb.row_bounds, world, rank, None)
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        allgather_rows(y, st.y2_all, sb.row_bounds, world, rank, None)
    torch.cuda.synchronize()
    tg = torch.tensor([(time.perf_counter() - t1) / 10], device=device, dtype=torch.float64)
    dist.all_reduce(tg, op=dist.ReduceOp.MAX)
    for p in params:
        p.grad = None
    ms = 1e3 * elapsed / steps
    recv = sb.gather_bytes - n_local * w["D"] * 4
    # per-rank step roofline with SURVEY 8d's byte model, de-dup aware: this rank's RGCN bytes on ITS distinct snapshots + the GRU
    # bytes of ITS windows' node visits + the exchanged rows once in, once out
    D = w["D"]
    n_gru = int(sb.program.n_total)
    rank_bytes = sb.n_edge_visits_local * 2 * (12 * D + 24) + n_local * 2 * (20 * D + 16) + n_gru * (32 * D + 4)
    ag_ms = 1e3 * float(tg.item())
    return dict(value=sb.n_edge_visits_global * steps / elapsed, unit="edges/s", ms_per_step=ms,
                global_windows=len(all_targets), rccl_ranks=dist.get_world_size(), launch=("3 hip graphs + eager collectives" if st.graphs is not None else "eager"),
                edge_visits_per_step_global=sb.n_edge_visits_global, distinct_edges_this_rank=sb.n_edge_visits_local,
                allgather_bytes_received_per_rank=recv, allgather_ms=ag_ms,
                allgather_gbps_per_rank=(recv / (ag_ms * 1e-3) / 1e9) if ag_ms > 0 and recv > 0 else None,
                per_rank_roofline=dict(bound="hbm", algorithmic_bytes=rank_bytes, achieved=rank_bytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                                       unit="GB/s", frac=rank_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
                parallelism="distinct snapshots/%d (edge-balanced) + direct all-gather(node states) + window-sharded GRU chain + grad all-reduce" % world)


def shbm_main(a, lib, device):
    """`--workload S-hbm`: the edge kernels ALONE at SURVEY 8d's full S-hbm snapshot size (2^20 nodes, 2^24 edges): ONE RGCN layer,
    forward + backward, over ONE snapshot whose 839-MB node matrix is far beyond the 256-MB Infinity Cache.  (The whole window path
    in this regime -- both layers, the GRU chain, every weight gradient -- is `--workload S-hbm-window` / `extra.hbm_window` at
    2^18 nodes / 2^22 edges per snapshot; the full-size window would hold 29 x ~17 node-row tensors of 839 MB = ~410 GB.)  `value` = edges processed per second (fwd + bwd of the layer); `roofline` = the forward aggregation kernel's
    ALGORITHMIC bytes (E (row + 8) + n row) over its HIP-event time against 8 TB/s, with the rocprofv3 PMC traffic of the same
    command (profiles/pmc_traffic_shbm*.json: FETCH_SIZE / WRITE_SIZE passes) beside it -- where the byte model exceeds the peak
    (Zipf hub rows are cache hits) `frac_from_counters` is the honest fraction."""
    from temp_amd import _lib, synthetic
    from temp_amd import backend as TB
    n, E, R, D, B = 1 << a.shbm_log2_nodes, 1 << a.shbm_log2_edges, a.shbm_relations, 200, 100
    t0 = time.perf_counter()
    g = synthetic.make_snapshots(n, R, E, n, 1, seed=0)[0]
    dg = g.device_graph(device, 2 * R)
    prep = time.perf_counter() - t0
    be = TB.get_backend()
    gen = torch.Generator(device="cpu").manual_seed(2)
    S = D // B
    w = (torch.rand(2 * R, B * S * S, generator=gen) - 0.5).to(device)
    lw = ((torch.rand(D, D, generator=gen) - 0.5) * 0.2).to(device)
    h = torch.randn(n, D, device=device)
    gy = torch.randn(n, D, device=device)

    def step():
        out = be.rgcn_fwd(dg, h, None, w, lw, None, B, 1)
        be.rgcn_bwd(dg, h, out, gy, w, lw, False, B, 1)

    for _ in range(max(a.warmup, 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tr = traced_steps(step, max(a.trace_steps, 1), lib) if a.trace_steps > 0 else {}
    row = 4 * D
    alg = {"k_rgcn_agg<fwd>": E * (row + 8) + n * row, "k_rgcn_agg<dx>": E * (row + 12) + n * row, "k_rgcn_dw": E * (2 * row + 12),
           "k_gemm_panel<loop_fwd>": 3 * n * row, "k_gemm_panel<loop_dx>": 3 * n * row, "k_relu_bwd": 3 * n * row}
    pmc_path = os.path.join(REPO, "profiles", "pmc_traffic_shbm%s.json" % ("" if R == 230 else "_rel%d" % R))
    per_kernel = json.load(open(pmc_path)).get("kernels", {}) if os.path.exists(pmc_path) else {}

    def traffic_of(trace_name):
        """HBM-side bytes per launch (PMC) of the kernel behind a trace id: k_rgcn_agg*<S, MODE, ...> with MODE 0 = fwd, 1 = d/dh."""
        import re
        for k, v in per_kernel.items():
            m = re.match(r"(k_rgcn_agg\w*)<\s*\d+\s*,\s*(\d)", k)
            if m and trace_name == ("k_rgcn_agg<fwd>" if m.group(2) == "0" else "k_rgcn_agg<dx>"):
                return v["traffic_bytes_per_launch"]
            if trace_name == "k_rgcn_dw" and k.startswith("k_rgcn_dw"):
                return v["traffic_bytes_per_launch"]
        return None

    kernels = {}
    for k, v in sorted(tr.items(), key=lambda kv: -kv[1]["ms_per_step"]):
        e = dict(launches_per_step=v["launches_per_step"], avg_ms=v["avg_ms"])
        if k in alg:
            e["algorithmic_bytes"] = alg[k]
            e["algorithmic_GBps"] = alg[k] / (v["avg_ms"] * 1e-3) / 1e9
            e["frac_of_8TBps_algorithmic"] = e["algorithmic_GBps"] / HBM_PEAK_GBS
        tb = traffic_of(k)
        if tb is not None:
            e["traffic_bytes"] = tb
            e["frac_of_8TBps_from_counters"] = tb / (v["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        kernels[k] = e
    roof = None
    dom = "k_rgcn_agg<fwd>"
    if dom in tr:
        ach = alg[dom] / (tr[dom]["avg_ms"] * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic_of(dom),
                    avg_launch_ms=tr[dom]["avg_ms"], algorithmic_per_launch=alg[dom])
        if roof["traffic"] is not None:
            roof["traffic_unit"] = "bytes/launch (rocprofv3 PMC, %s)" % os.path.relpath(pmc_path, REPO)
            roof["frac_from_counters"] = roof["traffic"] / (tr[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    out = dict(metric="edges/sec (fwd+bwd) of ONE RGCN layer over one S-hbm snapshot (edge kernels, HBM regime)", value=E * a.steps / elapsed, unit="edges/s",
               n_gpus=1, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f32", data="synthetic",
               config=dict(workload="S-hbm", nodes=n, edges=E, relations=R, embed=D, n_bases=B, host_prepare_s=prep,
                           what="one RGCN layer fwd+bwd on one snapshot (the full window in this regime: --workload S-hbm-window)"),
               roofline=roof, kernels=kernels, cpu_baseline=None)
    emit(out)


class GraphStep:
    """One forward+backward of `run()` with upstream gradient = ones, captured once into a HIP graph and replayed (eager when the
    capture fails or graph=False).  `grads` = the gradient tensors every replay writes."""

    def __init__(self, run, params, graph=True):
        self.run, self.params = run, params
        self._ones = {}
        self.graph = self.grads = None
        if graph:
            self._capture()

    def _backward_ones(self, out):
        # handed to backward() as a resident tensor: `.sum().backward()` computes the same gradients but spends a reduction, two
        # fills and -- because autograd expands the scalar's gradient with stride 0 -- one contiguous copy per consumer
        key = (tuple(out.shape), out.dtype)
        g = self._ones.get(key)
        if g is None:
            g = self._ones[key] = torch.ones_like(out)
        out.backward(g)

    def eager(self):
        for p in self.params:
            p.grad = None
        self._backward_ones(self.run())

    def _capture(self):
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.eager()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for p in self.params:
                p.grad = None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):     # RCCL's watchdog thread polls events meanwhile
                self._backward_ones(self.run())
            torch.cuda.synchronize()
            self.graph, self.grads = g, [p.grad for p in self.params]
        except Exception as e:                      # capture is an optimisation only
            print("bench: HIP graph capture failed (%s: %s); running eagerly" % (type(e).__name__, e), file=sys.stderr)
            self.graph = self.grads = None

    def __call__(self):
        if self.graph is None:
            return self.eager()
        self.graph.replay()

    def time(self, steps, warmup):
        for _ in range(warmup):
            self()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps


def launches_of(step_fn, lib):
    tr = traced_steps(step_fn, 1, lib)
    top = {k: dict(ms=round(v["ms_per_step"], 4), launches=int(round(v["launches_per_step"]))) for k, v in sorted(tr.items(), key=lambda kv: -kv[1]["ms_per_step"])[:6]}
    return int(round(sum(v["launches_per_step"] for v in tr.values()))), sum(v["ms_per_step"] for v in tr.values()), top

def other_config(name, a, device, lib, steps):
    """BASELINE.json configs 1-3 at their own sizes, each one short timed loop of the TRAINING step
    (encoder + all-entity pass + ComplEx loss, forward + backward) on one resident batch."""
    from temp_amd import synthetic
    from temp_amd.sampling import CorruptTriples
    if name == "config1_static":                 # ICEWS14 SRGCN, seq_len 1 (baselines/StaticRGCN.py:36-89)
        from temp_amd.static_rgcn import StaticRGCN
        w2 = synthetic.workload("S-icews14", seed=0)
        args = make_args(w2, "SRGCN")
        torch.manual_seed(1)
        m2 = StaticRGCN(args, w2["num_ents"], w2["num_rels"], w2["snapshots"], w2["snapshots"], w2["snapshots"]).to(device)
        m2.sample_rng = np.random.default_rng(2)
        m2.corrupter = CorruptTriples(m2.args, w2["snapshots"], seed=5)
        wb2 = m2.prepare(synthetic.default_targets(w2["num_times"], w2["L"], w2["bsz"], 3))
        edge_visits = int(wb2.n_edge_visits)
        with torch.no_grad():                                                              # one draw of negatives, then fixed (no autograd
            m2.run_loss(wb2)                                                               # graph may outlive this call: see GraphStep)
        fixed_cand = m2._last_plan[1]
        st = GraphStep(lambda: m2.run_loss(wb2, fixed_cand), [p for p in m2.parameters()], graph=not a.no_graph)
        what = ("StaticRGCN (2-layer RGCN, bias, ReLU; 50 % target edges), S-icews14 shape, bsz 8, seq_len 1: encoder + all-entity pass + "
                "ComplEx loss on one prepared batch, fixed negatives")
    elif name == "config2_default_flags":        # ICEWS14 GRRGCN seq_len 8, the reference's default flags (utils/args.py:38)
        from temp_amd.dynamic_rgcn import DynamicRGCN
        w2 = synthetic.workload("S-icews14", seed=0)
        args = make_args(w2, "GRRGCN")
        args.rec_only_last_layer = False
        torch.manual_seed(1)
        m2 = DynamicRGCN(args, w2["num_ents"], w2["num_rels"], w2["snapshots"], w2["snapshots"], w2["snapshots"]).to(device)
        m2.sample_rng = np.random.default_rng(2)
        m2.corrupter = CorruptTriples(m2.args, w2["snapshots"], seed=5)
        wb2 = m2.prepare(synthetic.default_targets(w2["num_times"], w2["L"], w2["bsz"], 3), w2["L"], True)
        fixed = [tuple(x.to(device) for x in smp) for smp in m2.draw_samples(wb2)]
        edge_visits = int(wb2.n_edge_visits)
        st = GraphStep(lambda: m2.run_loss(wb2, fixed), [p for p in m2.parameters()], graph=not a.no_graph)
        what = ("DynamicRGCN / GRRGCN, rec_only_last_layer=False (BOTH layers recurrent: the position loop as one autograd node, "
                "rec_stack.py), L=8, bsz 8, S-icews14 shape, encoder + loss, fixed negatives")
    else:                                        # ICEWS05-15 BiGRRGCN seq_len 15 --rec-only-last-layer --post-ensemble
        from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN
        w2 = synthetic.workload("S-icews0515", seed=0)
        args = make_args(w2, "BiGRRGCN")
        args.post_ensemble = True
        torch.manual_seed(1)
        m2 = PostEnsembleBiDynamicRGCN(args, w2["num_ents"], w2["num_rels"], w2["snapshots"], w2["snapshots"], w2["snapshots"]).to(device)
        m2.sample_rng = np.random.default_rng(2)
        m2.corrupter = CorruptTriples(m2.args, w2["snapshots"], seed=5)
        wb2 = m2.prepare(synthetic.default_targets(w2["num_times"], w2["L"], w2["bsz"], 3), w2["L"], True)
        fixed = [tuple(x.to(device) for x in smp) for smp in m2.draw_samples(wb2)]
        # the mixing weights come from frequency tables of utils/DropEdge.py (out of scope: the caller supplies them): 0.5 each
        wts = [(torch.full((smp[0].shape[0], 1), 0.5, device=device), torch.full((smp[0].shape[0], 1), 0.5, device=device)) for smp in fixed]
        edge_visits = int(wb2.n_edge_visits)
        st = GraphStep(lambda: m2.run_loss(wb2, fixed, wts), [p for p in m2.parameters()], graph=not a.no_graph)
        what = ("PostEnsembleBiDynamicRGCN (BiGRRGCN --rec-only-last-layer --post-ensemble), L=15, bsz 8, S-icews0515 shape, encoder "
                "(local + temporal streams) + ensemble loss, fixed negatives, mixing weights 0.5")
    n = max(10, min(steps, 30))
    ms = st.time(n, 3)
    nl, kms, top = launches_of(st.eager, lib)
    return dict(what=what, ms_per_step=ms, edges_per_s=edge_visits / (ms * 1e-3), edge_visits_per_step=edge_visits,
                launch="hip-graph replay" if st.graph is not None else "eager", steps=n, launches_per_step=nl, kernel_ms_per_step=kms,
                top_kernels=top)



def extra_child_main(a):
    claim_stdout()
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    from temp_amd import _lib
    lib = _lib.load()
    out = other_config(a.extra_child, a, torch.device("cuda", 0), lib, max(20, min(a.steps, 100)))
    emit(out)


def extra_measurements(a, w, model, wb, targets, device, lib):
    """Secondary measurements of the default run (rank 0, one GPU), each a short object; a failure is reported in its object and
    never costs the headline line."""
    from temp_amd import synthetic
    out = {}
    steps, warm = max(20, min(a.steps, 100)), 5
    params = [p for p in model.parameters()]

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:
            out[name] = dict(error="%s: %s" % (type(e).__name__, e))
        out[name]["wall_s"] = round(time.perf_counter() - t0, 2)

    def with_loss():
        from temp_amd.sampling import CorruptTriples
        model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
        fixed = [tuple(x.to(device) for x in smp) for smp in model.draw_samples(wb)]
        st = GraphStep(lambda: model.run_loss(wb, fixed), params, graph=not a.no_graph)
        ms = st.time(steps, warm)
        return dict(what="encoder + all-entity pass + ComplEx scores + cross-entropy, negative_rate 500, fixed negatives", ms_per_step=ms,
                    edges_per_s=wb.n_edge_visits / (ms * 1e-3), launch="hip-graph replay" if st.graph is not None else "eager", steps=steps)

    def attention():
        m2 = build_model(w, device, "attention")
        m2.sample_rng = np.random.default_rng(2)
        wb2 = m2.prepare(targets, w["L"], train=True)
        st = GraphStep(lambda: m2.run(wb2)[0], [p for p in m2.parameters()], graph=not a.no_graph)
        ms = st.time(steps, warm)
        return dict(what="BiSelfAttentionRGCN (config 5), same windows", ms_per_step=ms, edges_per_s=wb2.n_edge_visits / (ms * 1e-3),
                    launch="hip-graph replay" if st.graph is not None else "eager", steps=steps)

    def sharded():
        import torch.distributed as dist
        own = not dist.is_initialized()
        if own:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1)
        try:
            r = measure_sharded(model, w, 1, 0, device, steps, warm, dist, graphs=not a.no_graph)
        finally:
            if own:
                dist.destroy_process_group()
        return dict(what="north-star snapshot-sharded step on ONE RCCL rank (three HIP graphs around the two exchanges + grad all-reduce)",
                    ms_per_step=r["ms_per_step"], edges_per_s=r["value"], launch=r["launch"], rccl_ranks=r["rccl_ranks"], steps=steps)

    def default_dropout():
        """The training step with the reference's DEFAULT dropout (0.1 on the self-loop message, utils/args.py:17; SURVEY 8d measures
        at 0): every visit and every window's isolated pass draws its own mask, so no snapshot is shared between windows and the step
        is not captured (the masks' seeds are drawn per call) -- eager launches, beside the same step without dropout, also eager."""
        from temp_amd.sampling import CorruptTriples
        m2 = build_model(w, device, dropout=0.1)
        m2.train()
        m2.sample_rng = np.random.default_rng(2)
        m2.corrupter = CorruptTriples(m2.args, w["snapshots"], seed=5)
        wb2 = m2.prepare(targets, w["L"], train=True)
        fixed2 = [tuple(x.to(device) for x in smp) for smp in m2.draw_samples(wb2)]
        st2 = GraphStep(lambda: m2.run_loss(wb2, fixed2), [p for p in m2.parameters()], graph=False)
        ms2 = st2.time(steps, warm)
        model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
        fixed0 = [tuple(x.to(device) for x in smp) for smp in model.draw_samples(wb)]
        st0 = GraphStep(lambda: model.run_loss(wb, fixed0), params, graph=False)
        ms0 = st0.time(steps, warm)
        return dict(what="encoder + all-entity pass + loss, fwd+bwd, dropout 0.1 (the reference's default): per-visit masks, nothing shared between "
                         "windows, eager launches", ms_per_step=ms2, edges_per_s=wb2.n_edge_visits / (ms2 * 1e-3), launch="eager",
                    rgcn_node_rows=int(wb2.n_nodes_distinct), node_visits=int(wb2.n_node_visits),
                    no_dropout_eager_ms_per_step=ms0, steps=steps)

    guarded("with_loss", with_loss)
    guarded("default_dropout", default_dropout)
    guarded("attention", attention)
    guarded("sharded_1rank", sharded)
    def config_in_child(name):
        """The other BASELINE configs build their own models and capture their own HIP graphs: each runs in a child process, so
        that nothing it does (a capture that goes wrong ends the process, not with an exception) can cost the headline line."""
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--extra-child", name, "--steps", str(steps)] + (["--no-graph"] if a.no_graph else [])
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return dict(error="child process ended with code %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else ""))
        return json.loads(lines[-1])

    for cfg_name in ("config1_static", "config2_default_flags", "config3_post_ensemble"):
        guarded(cfg_name, lambda cfg_name=cfg_name: config_in_child(cfg_name))
    if a.hbm_window_log2_nodes > 0:
        guarded("hbm_window", lambda: hbm_window(a, device, lib))
    return out


def hbm_window(a, device, lib):
    """The WHOLE hot path in the HBM regime (BASELINE.md section 3, S-hbm row scaled to one GPU): BiGRRGCN, L = 15, bsz = 1 --
    29 snapshot visits of 2^k nodes / 2^(k+4) edges each (k = 18: a 210-MB node matrix per visit, ~100 GB of activations, far beyond
    the 256-MB Infinity Cache), 230 relations, through the same batched step as the headline (two RGCN layers over the union graph,
    input-gate GEMM, window-chain kernels, all weight gradients).  roofline = SURVEY 8d's per-visit byte model against 8 TB/s."""
    from temp_amd import synthetic
    k = a.hbm_window_log2_nodes
    N, E, R, D, B, L = 1 << k, 1 << (k + 4), a.shbm_relations, 200, 100, 15
    t0 = time.perf_counter()
    snaps = synthetic.make_snapshots(N, R, E, N, 2 * L - 1, seed=0)
    w = dict(name="S-hbm-window", num_ents=N, num_rels=R, edges_per_snap=E, nodes_per_snap=N, num_times=2 * L - 1, D=D, B=B, L=L, bsz=1,
             module="BiGRRGCN", snapshots=snaps)
    gen_s = time.perf_counter() - t0
    model = build_model(w, device)
    model.sample_rng = np.random.default_rng(2)
    t0 = time.perf_counter()
    from temp_amd import snapshot as S_
    S_.prepack_parallel(list(snaps.values()), 2 * R, threads=8)       # (the 29 snapshots' sorted views: host work, several at a time)
    wb = model.prepare([L - 1], L, train=True)
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t0
    params = [p for p in model.parameters()]
    st = GraphStep(lambda: model.run(wb)[0], params, graph=False)      # (a 0.2-s step: launch overhead is nothing here)
    steps = max(2, a.hbm_window_steps)
    ms = st.time(steps, 1)
    tr = traced_steps(st.eager, 1, lib)
    total = sum(v["ms_per_step"] for v in tr.values())
    top = sorted(tr.items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]
    n_over_e = wb.n_node_visits / max(wb.n_edge_visits, 1)
    bpe = 2 * (12 * D + 24) + n_over_e * (2 * (20 * D + 16) + 32 * D + 4)
    ach = wb.n_edge_visits * bpe / (ms * 1e-3) / 1e9
    mem = torch.cuda.max_memory_allocated(device) / 2 ** 30
    edge_visits, node_visits = int(wb.n_edge_visits), int(wb.n_node_visits)
    # (counter passes exist for the window sizes of the rounds that took them: r04 at 2^18 nodes, r05 / r06 at 2^19)
    pmc_path = os.path.join(REPO, "profiles", "none")
    for rnd in ({19: ("r06", "r05"), 18: ("r04",)}.get(k, ())):          # newest counter pass taken at this window size
        cand = os.path.join(REPO, "profiles", "%s_pmc_traffic_hbm_window.json" % rnd)
        if os.path.exists(cand):
            pmc_path = cand
            break
    traffic = json.load(open(pmc_path)).get("step_traffic_bytes") if os.path.exists(pmc_path) else None
    del st, wb, model, snaps
    torch.cuda.empty_cache()
    return dict(what="BiGRRGCN --rec-only-last-layer, L=15, bsz=1, one full bi window (29 snapshot visits), encoder fwd+bwd, eager launches",
                nodes_per_snapshot=N, edges_per_snapshot=E, relations=R, embed=D, edge_visits_per_step=edge_visits, node_visits_per_step=node_visits, ms_per_step=ms,
                edges_per_s=ach * 1e9 / bpe, steps=steps,
                roofline=dict(bound="hbm", model="SURVEY 8d: 4848 + 14436 (n/E) bytes per snapshot-edge visit", bytes_per_edge_visit=bpe,
                              achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                              traffic=traffic, traffic_source=("static: profiles/%s (rocprofv3 --pmc passes of this workload, bytes per step)"
                                                               % os.path.basename(pmc_path) if traffic else None),
                              frac_from_counters=(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None),
                kernels={k_: dict(ms=v["ms_per_step"], launches=v["launches_per_step"], share=v["ms_per_step"] / total) for k_, v in top},
                traced_kernel_ms=total, peak_memory_gib=mem, host_generate_s=gen_s, host_prepare_s=prep_s)


def _cpu_quota():
    """CPUs this process may use at once: the cgroup's CPU quota (cpu.max of cgroup v2 / cfs_quota_us of v1), else the affinity
    mask's size.  (The GPU boxes show 256 hardware threads and a quota of 16: more than 16 busy threads are throttled, which is
    why 16 torch threads are the oracle's optimum there and why an `all 256 threads` run does not finish.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(float(q) / float(per)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(round(q / per))))
        except (OSError, ValueError):
            pass
    return n


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_JSON_OUT = None


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n, argv, port=None):
    """The command `python bench.py --gpus N ...` re-executes itself as when no launcher has set WORLD_SIZE: one process per GPU of
    this node under torch.distributed.run (the reference's axis: Lightning DDP, models/TKG_Module.py:162-179, launcher_2gpu.sh:8),
    rendezvous on 127.0.0.1 (a container hostname may not resolve).  The children see RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    and rank 0 alone prints the JSON line."""
    port = port or int(os.environ.get("TEMP_BENCH_MASTER_PORT", "0")) or _free_port()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    import subprocess
    cmd = launch_command(n, argv)
    sys.stderr.write("[bench] WORLD_SIZE unset and --gpus %d: launching %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)                  # the ranks inherit stdout: rank 0's JSON line is this process's one line


def rendezvous_only(a, rank, world):
    """--rendezvous-only: prove that the launch path forms an N-rank group (tests/test_bench_launch.py runs it without a GPU)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl" if gpu else "gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)], device="cuda" if gpu else "cpu")
    dist.all_reduce(t)
    if rank == 0:
        emit(dict(rendezvous=world, backend="nccl" if gpu else "gloo", rank_sum=t.item(), gpus=a.gpus))
    dist.barrier()
    dist.destroy_process_group()


def claim_stdout():
    """stdout carries ONE line, the result: libraries that write to file descriptor 1 on their own (RCCL prints a version banner
    when a communicator is created -- every multi-GPU run, and the one-rank sharded measurement of `extra`) are pointed at
    stderr for the rest of the process, and emit() writes the JSON line to the original stdout."""
    global _JSON_OUT
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    _JSON_OUT = os.fdopen(real, "w")


def emit(obj):
    f = _JSON_OUT if _JSON_OUT is not None else sys.stdout
    f.write(json.dumps(obj) + "\n")
    f.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (0.5 s of replays at the headline shape: boxes differ by a few per cent)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="S-gdelt")
    ap.add_argument("--trace-steps", type=int, default=3)
    ap.add_argument("--shard", choices=("both", "windows", "snapshots"), default="both",
                    help="N>1: 'windows' = each rank encodes its own windows (+ gradient all-reduce), the reference's DDP axis, HIP-graph "
                         "replay; 'snapshots' = the distinct snapshots of a global batch of bsz*N windows sharded across ranks with a "
                         "direct all-gather of per-snapshot node states before the recurrent chain (BASELINE north_star; three HIP graphs around the two exchanges); "
                         "'both' (default) = the headline `value` is the windows mode and the snapshot-sharded measurement rides along "
                         "under `north_star_sharded`")
    ap.add_argument("--sharded-timeout", type=int, default=240, help="N > 1, --shard both: seconds the snapshot-sharded measurement may take "
                                                                       "before the headline line is printed without it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch threads of the cpu_baseline leg (16 = the measured optimum on the 256-thread box)")
    ap.add_argument("--no-fp32-mfma-compare", action="store_true",
                    help="skip the second, short timing with every product on the fp32 MFMA kernels (config.fp32_mfma_ms_per_step)")
    ap.add_argument("--train-loop-steps", type=int, default=100,
                    help="also time this many steps of a real training loop (new batch every step: host prepare, fresh negatives, loss, "
                         "backward, Adam, eager launches) and report it under config.train_loop, after the timed region (never the "
                         "headline); pass 0 when profiling, so that a rocprofv3 summary holds the headline step's kernels only")
    ap.add_argument("--with-loss", action="store_true",
                    help="also run the all-entity pass + scorer + cross-entropy (negative_rate 500, fixed negatives) in the step "
                         "(reported separately from the headline encoder-only metric, SURVEY 8d)")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel trace table to stderr")
    ap.add_argument("--encoder", choices=("gru", "attention"), default="gru",
                    help="gru: the headline RGCN+GRU window models; attention: SelfAttentionRGCN / BiSelfAttentionRGCN (config 5, "
                         "secondary measurement: no cpu_baseline leg)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying the step as a captured HIP graph")
    ap.add_argument("--shbm-relations", type=int, default=230, help="--workload S-hbm: relations (230 = SURVEY's, 20 = GDELT's: the weight table then fits LDS)")
    ap.add_argument("--shbm-log2-nodes", type=int, default=20)
    ap.add_argument("--shbm-log2-edges", type=int, default=24)
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (`extra`: loss, attention, sharded on one rank, HBM-regime window)")
    ap.add_argument("--hbm-window-log2-nodes", type=int, default=19,
                    help="extra.hbm_window: nodes per snapshot = 2^k, edges = 2^(k+4), bi L=15 bsz=1 (0: skip; 19 = the largest window one "
                         "288-GB GPU holds: 166 GiB peak, ~85 s of host generation + planning; 18: 83 GiB, ~35 s; 20 does not fit)")
    ap.add_argument("--hbm-window-steps", type=int, default=3)
    ap.add_argument("--cpu-probe", action="store_true", help="(internal) child process of the all-cores CPU probe: no GPU")
    ap.add_argument("--cpu-probe-seconds", type=float, default=0.0, help="(internal) --cpu-probe: whole batches for this many seconds (0: one window)")
    ap.add_argument("--cpu-probe-rank", type=int, default=0, help="(internal) --cpu-probe: which targets this process takes")
    ap.add_argument("--extra-child", default="", help="(internal) child process of one `extra.config*` measurement: prints its JSON object")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="(test hook) join the process group (nccl with a GPU, gloo without), all-reduce the rank ids, print one line and exit")
    a = ap.parse_args()
    if a.cpu_probe:
        return cpu_probe_main(a)
    if a.extra_child:
        return extra_child_main(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started plainly: start the N ranks ourselves (one process per GPU on this node)
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))
    claim_stdout()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.rendezvous_only:
        return rendezvous_only(a, rank, world)
    assert world == a.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, a.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    # one process per GPU.  TEMP_BENCH_DIST_BACKEND=gloo is a test hook for one-GPU boxes: RCCL refuses two ranks on one device, gloo
    # moves the same device tensors through host staging, so `--gpus 2` walks the whole multi-rank control flow (every collective of
    # every rank, in order) on ONE GPU -- a functional check of the launch path, never a measurement.
    dist_backend = os.environ.get("TEMP_BENCH_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    assert local_rank < n_dev or dist_backend != "nccl", "rank %d has no GPU of its own (%d visible)" % (local_rank, n_dev)
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("TEMP_BENCH_FORCE_DIST") == "1":       # FORCE_DIST: exercise the RCCL path on one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(dist_backend, rank=rank, world_size=world)

    from temp_amd import _lib, synthetic
    from temp_amd import backend as TB
    lib = _lib.load()
    assert TB.get_backend().name == "hip"
    global MFMA_MODE, CPU_THREADS
    CPU_THREADS = max(1, a.cpu_threads)
    MFMA_MODE = ("f16x2" if lib.temp_get_option(OPT_MFMA_F16X2) else "bf16x3") if lib.temp_get_option(OPT_MFMA_BF16X3) else "f32"

    if a.workload == "S-hbm-window":                 # the HBM-regime window on its own (profiling runs): same object as extra.hbm_window
        if rank == 0:
            r = hbm_window(a, device, lib)
            emit(dict(metric="edges/sec (fwd+bwd) RGCN+GRU seq_len=15, one full bi window in the HBM regime", value=r["edges_per_s"], unit="edges/s",
                                  n_gpus=1, steps=r["steps"], warmup=1, ms_per_step=r["ms_per_step"], higher_is_better=True, scaling="weak", vs_baseline=None,
                                  dtype="f32", data="synthetic", config=dict(workload="S-hbm-window", **{k: r[k] for k in ("nodes_per_snapshot", "edges_per_snapshot", "relations", "embed", "edge_visits_per_step", "what")}),
                                  roofline=r["roofline"], kernels=r["kernels"], cpu_baseline=None, detail=r))
        if dist is not None:
            dist.destroy_process_group()
        return
    if a.workload == "S-hbm":
        if rank == 0:
            shbm_main(a, lib, device)
        if dist is not None:
            dist.destroy_process_group()
        return
    w = synthetic.workload(a.workload, seed=0)
    model = build_model(w, device, a.encoder)
    if a.encoder == "attention":
        a.no_cpu_baseline = True
    bi = w["module"].startswith("Bi")
    from temp_amd.dist import SnapshotShardedEncoder, allreduce_gradients
    sharded = dist is not None and a.shard == "snapshots"
    ns_result = None
    targets = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], rank)
    params = [p for p in model.parameters()]
    t0 = time.perf_counter()
    if sharded:
        # one global batch of bsz*world windows, identical on every rank (same seed => same target subsample)
        all_targets = [t for r in range(world) for t in synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r)]
        model.sample_rng = np.random.default_rng(2)
        enc = SnapshotShardedEncoder(model)
        wb = enc.prepare(all_targets, w["L"], train=True)
        wb.n_edge_visits = wb.n_edge_visits_global / world       # per-rank share of the global step
        wb.n_node_visits = 0
        run = lambda: enc.run(wb)
        from temp_amd.dist import ShardedStep
        sharded_step = ShardedStep(enc, wb, params, graphs=not a.no_graph, average=False, force_allreduce=True)
    else:
        model.sample_rng = np.random.default_rng(2 + rank)
        wb = model.prepare(targets, w["L"], train=True)
        wb.n_table_rows = w["num_ents"]
        if a.with_loss:
            from temp_amd.sampling import CorruptTriples
            model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5 + rank)
            fixed = [tuple(x.to(device) for x in smp) for smp in model.draw_samples(wb)]
            run = lambda: model.run_loss(wb, fixed)
        else:
            run = lambda: model.run(wb)[0]
    torch.cuda.synchronize()
    prepare_s = time.perf_counter() - t0

    # The window batch is static, so the whole forward+backward of a step is captured once into a HIP graph and replayed
    # (GraphStep); the gradient all-reduce stays outside the graph.
    gs = None
    if not sharded:
        gs = GraphStep(run, params, graph=not a.no_graph)
    graph, graph_grads = (gs.graph, gs.grads) if gs is not None else (None, None)
    step_eager_local = gs.eager if gs is not None else None

    def step_eager():
        step_eager_local()
        if dist is not None:
            allreduce_gradients(params, world, average=not sharded)

    def capture():
        g2 = GraphStep(run, params, graph=True)
        return g2.graph, g2.grads

    def step():
        if sharded:
            return sharded_step.step()
        if graph is None:
            return step_eager()
        graph.replay()
        if dist is not None:
            allreduce_gradients(params, world, average=True, grads=graph_grads)

    for _ in range(a.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    edges = float(wb.n_edge_visits)
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        e = torch.tensor([edges], device=device, dtype=torch.float64)
        dist.all_reduce(e)
        edges = float(e.item())
    value = edges * a.steps / elapsed
    if dist is not None and a.shard == "both" and a.encoder == "gru" and not a.with_loss:
        # The snapshot-sharded measurement (north_star's variant) runs AFTER the headline, under a watchdog: its point-to-point
        # exchange has never met more than one GPU (one-GPU boxes), and a rank stuck in it must not cost the headline line.  When
        # the timer fires, every rank is stuck in the same place: rank 0 prints the line with what it has and all ranks leave.
        import threading
        finished = threading.Event()

        def bail():
            if finished.is_set():
                return
            if rank == 0:
                emit(dict(metric="edges/sec (fwd+bwd) RGCN+GRU seq_len=%d" % w["L"], value=value, unit="edges/s", n_gpus=world, steps=a.steps,
                          warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
                          dtype="f32", data="synthetic",
                          config=dict(workload=w["name"], encoder=w["module"], rec_only_last_layer=True, seq_len=w["L"], windows_per_gpu=w["bsz"],
                                      embed=w["D"], n_bases=w["B"], parallelism="dp%d(windows)+grad-allreduce" % world,
                                      rccl_ranks=dist.get_world_size(), launch="hip-graph replay" if graph is not None else "eager"),
                          roofline=None, cpu_baseline=None,
                          north_star_sharded=dict(error="the snapshot-sharded measurement did not finish within %d s (watchdog); the headline "
                                                        "above was measured before it" % a.sharded_timeout)))
            os._exit(0)

        timer = threading.Timer(a.sharded_timeout, bail)
        timer.daemon = True
        timer.start()
        try:
            ns_result = measure_sharded(model, w, world, rank, device, a.steps, a.warmup, dist, graphs=not a.no_graph)
        except Exception as e:                      # never lose the headline line to the secondary measurement
            ns_result = dict(error="%s: %s" % (type(e).__name__, e))
        finished.set()
        timer.cancel()
    if dist is not None and rank != 0:
        # Everything below is rank 0's own work (kernel trace, CPU baseline, the line) and contains NO collective: the other ranks
        # leave the group here instead of waiting minutes inside one.
        dist.destroy_process_group()
        return

    roof = None
    cpu = None
    if rank == 0 and a.trace_steps > 0 and not sharded:
        tr = traced_steps(step_eager_local, a.trace_steps, lib)  # per-kernel events need eager launches; this rank's kernels only (no all-reduce: the peers have left)
        costs = algorithmic_costs(wb, w["D"], bi, w["D"] // w["B"])
        total_ms = sum(v["ms_per_step"] for v in tr.values())
        dom = max(tr, key=lambda k: tr[k]["ms_per_step"])
        if a.kernel_table:
            for k in sorted(tr, key=lambda k: -tr[k]["ms_per_step"]):
                v = tr[k]
                cst = costs.get(k, {})
                gbs = cst.get("bytes", 0) / (v["ms_per_step"] * 1e-3) / 1e9 if v["ms_per_step"] else 0
                tfs = cst.get("flops", 0) / (v["ms_per_step"] * 1e-3) / 1e12 if v["ms_per_step"] else 0
                print("%-28s launches/step %6.1f  ms/step %8.3f (%4.1f%%)  avg %8.4f ms  alg %7.1f GB/s %6.1f TF/s"
                      % (k, v["launches_per_step"], v["ms_per_step"], 100 * v["ms_per_step"] / total_ms, v["avg_ms"], gbs, tfs),
                      file=sys.stderr)
        # HBM-side bytes per launch: measured separately with rocprofv3 --pmc (FETCH_SIZE / WRITE_SIZE in their own passes,
        # tools/pmc_summary.py) on this same workload and committed under profiles/ -- counters cannot be collected from inside this
        # process, so these are STATIC numbers and say so.
        pmc_path = profile_file("pmc_traffic.json")
        pmc = json.load(open(pmc_path)) if (os.path.exists(pmc_path) and a.workload == "S-gdelt" and a.encoder == "gru" and not a.with_loss) else None

        def pmc_traffic(name):
            """PMC bytes per launch of the rocprof kernels behind a trace name (k_rgcn_agg<fwd|dx> = k_rgcn_agg*<S, MODE 0|1, ...>)."""
            import re
            if not pmc:
                return None
            hits = []
            for k, v in pmc.get("kernels", {}).items():
                m = re.match(r"(k_rgcn_agg\w*)<\s*\d+\s*,\s*(\d)", k)
                if name in ("k_rgcn_agg<fwd>", "k_rgcn_agg<dx>"):
                    if m and name == ("k_rgcn_agg<fwd>" if m.group(2) == "0" else "k_rgcn_agg<dx>"):
                        hits.append(v)
                elif name == "k_rgcn_dw":
                    if k.startswith("k_rgcn_dw"):
                        hits.append(v)
                elif k == name or k.startswith(name + "<") or k.startswith(name + "_multi<") or k.startswith(name + "_hx<"):       # (_hx: the f16 kernels of round 6 under the family's trace name)
                    hits.append(v)
            n_l = sum(h["launches"] for h in hits)
            return sum(h["traffic_bytes_per_launch"] * h["launches"] for h in hits) / n_l if n_l else None

        def kernel_roof(name):
            """Roofline entry of one traced kernel: algorithmic flops (MFMA kernels) or SURVEY 8d's bytes (the rest) per launch over
            its HIP-event average, against the pipe that bounds it."""
            cst = costs.get(name, dict(bytes=0, flops=0))
            sec = tr[name]["ms_per_step"] * 1e-3
            if name.startswith(MFMA_KERNELS) and cst["flops"]:
                ach = cst["flops"] / sec / 1e12                     # algorithmic (fp32) flops
                if MFMA_MODE == "f16x2" and name.startswith(HX_KERNELS):
                    peak = MFMA_BF16_PEAK_TFLOPS / 3.0
                    r = dict(bound="mfma", kernel=name, achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, traffic=None,
                             pipe="f16 MFMA, three products per fp32 product (scaled 2-way operand split): peak = %.0f / 3" % MFMA_BF16_PEAK_TFLOPS,
                             executed_f16_tflops=3.0 * ach, frac_of_fp32_mfma_peak=ach / MFMA_F32_PEAK_TFLOPS,
                             frac_of_bf16x3_roof=ach / (MFMA_BF16_PEAK_TFLOPS / 6.0))
                elif MFMA_MODE in ("bf16x3", "f16x2") and name.startswith(BX_KERNELS):
                    peak = MFMA_BF16_PEAK_TFLOPS / 6.0
                    r = dict(bound="mfma", kernel=name, achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, traffic=None,
                             pipe="bf16 MFMA, six products per fp32 product (exact 3-way operand split): peak = %.0f / 6" % MFMA_BF16_PEAK_TFLOPS,
                             executed_bf16_tflops=6.0 * ach, frac_of_fp32_mfma_peak=ach / MFMA_F32_PEAK_TFLOPS)
                else:
                    r = dict(bound="mfma", kernel=name, achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=ach / MFMA_F32_PEAK_TFLOPS, traffic=None, pipe="fp32 MFMA")
            else:
                ach = cst["bytes"] / sec / 1e9
                r = dict(bound="hbm", kernel=name, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None)
            tb = pmc_traffic(name)
            if tb is not None:
                r["traffic"] = tb
                r["traffic_unit"] = "bytes/launch"
                r["traffic_source"] = "static: profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not this run)" % os.path.basename(pmc_path)
            r["algorithmic_per_launch"] = cst["flops" if r["bound"] == "mfma" else "bytes"] / max(tr[name]["launches_per_step"], 1e-9)
            r.update(avg_launch_ms=tr[name]["avg_ms"], launches_per_step=tr[name]["launches_per_step"], share_of_kernel_time=tr[name]["ms_per_step"] / total_ms)
            if r["bound"] == "hbm" and r["traffic"] is not None and r["traffic"] < 0.5 * r["algorithmic_per_launch"]:
                r["bound"] = "lds-issue"
                r["frac_of_byte_model"] = r["frac"]
                r["frac_from_counters"] = (r["traffic"] / (tr[name]["avg_ms"] * 1e-3) / 1e9) / HBM_PEAK_GBS
                r["note"] = ("the byte model charges every edge a row from HBM; this kernel stages a member snapshot's rows in LDS once and the "
                             "counters see %.0f %% of the model's bytes: it is bound by the LDS walk's instruction issue (HISTORY.md section 3), "
                             "not by HBM -- read `frac` as work rate against the survey's model, `traffic` as what HBM saw"
                             % (100.0 * r["traffic"] / r["algorithmic_per_launch"]))
            return r

        # roofline.kernel = the LONGEST single launch among the MFMA kernels (a fixed rule: the entry cannot change its meaning by
        # which family happens to lead the time shares; an edge kernel whose counters see a fraction of the byte model is LDS-issue
        # bound and is listed under `others` with bound = "lds-issue" and frac_from_counters)
        mf = [k for k in tr if k.startswith(MFMA_KERNELS) and costs.get(k, {}).get("flops")]
        prim = max(mf, key=lambda k: tr[k]["avg_ms"]) if mf else dom
        roof = kernel_roof(prim)
        if roof["bound"] == "mfma":
            # the same launch against HBM with SURVEY 8d's bytes: a persistent chain kernel is bound by neither pipe alone
            roof["hbm_view"] = dict(algorithmic_bytes_per_launch=costs[prim]["bytes"] / max(tr[prim]["launches_per_step"], 1e-9),
                                    achieved_gbs=costs[prim]["bytes"] / (tr[prim]["ms_per_step"] * 1e-3) / 1e9,
                                    frac=costs[prim]["bytes"] / (tr[prim]["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
        roof["largest_share_kernel"] = dom
        roof["others"] = [{k: v for k, v in kernel_roof(n).items() if k not in ("pipe", "traffic_source", "traffic_unit")}
                          for n in sorted(tr, key=lambda k: -tr[k]["ms_per_step"])[:5] if n in costs and n != prim][:4]
        roof.update(traced_kernel_ms_per_step=total_ms,
                    timing="HIP events around every launch (library event trace on the launch stream) of %d EAGER steps run right after the "
                           "timed region (one more first step dropped, medians over the kept steps); inside the HIP-graph replays of the timed region the same kernel runs 5-10 %% faster "
                           "(rocprofv3, profiles/r06_bench_kernel_stats.md and r06_step_sequence.txt)" % a.trace_steps)
        # whole-step view against the HBM roofline with SURVEY 8d's byte model (2 RGCN layers + 1 GRU cell, fwd+bwd, fp32, int32 ids):
        #   per edge 2*(12D+24) B, per RGCN node row 2*(20D+16) B, per GRU row 32D+4 B.
        # (a) as the survey states it, per snapshot-edge VISIT (every visit pays its RGCN bytes), and
        # (b) for the work the kernels actually do: the RGCN layers run once per DISTINCT snapshot of the step (a snapshot shared by
        #     overlapping windows is computed once -- a result-identical restructuring), the GRU once per node visit.
        D = w["D"]
        n_over_e = wb.n_node_visits / max(wb.n_edge_visits, 1)
        bytes_per_edge = 2 * (12 * D + 24) + n_over_e * (2 * (20 * D + 16) + 32 * D + 4)
        steps_per_s = a.steps / elapsed
        E_d = getattr(wb, "n_edges_distinct", 0) or wb.n_edge_visits
        n_d = getattr(wb, "n_nodes_distinct", 0) or wb.n_node_visits
        n_gru = wb.n_node_visits + (wb.target.n_rows if (bi and hasattr(wb, "target")) else 0)
        bytes_dedup = E_d * 2 * (12 * D + 24) + n_d * 2 * (20 * D + 16) + n_gru * (32 * D + 4)
        flops_step = sum(c.get("flops", 0) for k, c in costs.items() if k in tr)
        roof["step_bytes_per_edge_visit"] = bytes_per_edge
        roof["step_frac_of_hbm_visit_model"] = wb.n_edge_visits * steps_per_s * bytes_per_edge / (HBM_PEAK_GBS * 1e9)
        roof["step_algorithmic_bytes"] = bytes_dedup
        roof["step_frac_of_hbm"] = bytes_dedup * steps_per_s / (HBM_PEAK_GBS * 1e9)
        roof["step_algorithmic_flops"] = flops_step
        roof["step_frac_of_mfma"] = flops_step * steps_per_s / (MFMA_F32_PEAK_TFLOPS * 1e12)
        if pmc and pmc.get("step_traffic_bytes"):
            roof["step_traffic_bytes"] = pmc["step_traffic_bytes"]
            roof["step_traffic_over_algorithmic"] = pmc["step_traffic_bytes"] / bytes_dedup
            roof["step_traffic_source"] = "static: profiles/%s" % os.path.basename(pmc_path)
    # a longer replay of the same graph when the timed region was short (the driver's 20 steps are 50 ms: box-to-box noise is larger
    # than the 1-3 % steps a round works on); reported beside the headline, never instead of it
    long_ms = None
    if rank == 0 and world == 1 and not sharded and a.steps < 200 and a.trace_steps > 0 and not a.no_graph:      # (not in counter / eager runs)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        tl = time.perf_counter()
        for _ in range(400):
            step()
        torch.cuda.synchronize()
        long_ms = 1e3 * (time.perf_counter() - tl) / 400
    if rank == 0 and not a.no_cpu_baseline:
        cpu = cpu_baseline(model, w, targets)
    extra = None
    if rank == 0 and world == 1 and not sharded and not a.no_extras and a.workload == "S-gdelt" and a.encoder == "gru" and not a.with_loss:
        extra = extra_measurements(a, w, model, wb, targets, device, lib)
    loop = None
    if rank == 0 and world == 1 and a.train_loop_steps > 0 and not sharded:
        try:
            loop = train_loop(model, w, a.train_loop_steps)
        except Exception as e:                      # an extra, never the headline
            print("bench: training-loop probe failed (%s: %s)" % (type(e).__name__, e), file=sys.stderr)

    fp32_ms = None
    if rank == 0 and world == 1 and not sharded and MFMA_MODE in ("bf16x3", "f16x2") and not a.no_fp32_mfma_compare:
        # the same step with every product on the fp32 MFMA kernels: one library switch, same process, same batch, graph re-captured
        try:
            lib.temp_set_option(OPT_MFMA_BF16X3, 0)
            g32, _ = capture() if graph is not None else (None, None)
            one = g32.replay if g32 is not None else step_eager
            n32 = min(a.steps, 50)
            for _ in range(min(a.warmup, 5)):
                one()
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            for _ in range(n32):
                one()
            torch.cuda.synchronize()
            fp32_ms = 1e3 * (time.perf_counter() - t32) / n32
            del g32
        except Exception as e:                      # an extra, never the headline
            print("bench: fp32-MFMA comparison run failed (%s: %s)" % (type(e).__name__, e), file=sys.stderr)
        finally:
            lib.temp_set_option(OPT_MFMA_BF16X3, 1)

    if rank == 0:
        out = dict(metric="edges/sec (fwd+bwd) RGCN+%s seq_len=%d%s" % ("GRU" if a.encoder == "gru" else "self-attention", w["L"], " + link-prediction loss" if a.with_loss else ""), value=value, unit="edges/s", n_gpus=world,
                   steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * elapsed / a.steps, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f32", data="synthetic",
                   value_note="edge VISITS per second (a visit = one snapshot at one window position of one window, BASELINE.md section 3); "
                              "value_distinct = edges the RGCN kernels actually process per second (snapshots shared by overlapping "
                              "windows are computed once)",
                   value_distinct=(value * getattr(wb, "n_edges_distinct", wb.n_edge_visits) / max(wb.n_edge_visits, 1)) if not sharded else None,
                   config=dict(workload=w["name"], encoder=w["module"] if a.encoder == "gru" else ("BiSARGCN" if bi else "SARGCN"), rec_only_last_layer=True, seq_len=w["L"],
                               windows_per_gpu=w["bsz"], embed=w["D"], n_bases=w["B"], entities=w["num_ents"],
                               relations=w["num_rels"], edges_per_snapshot=w["edges_per_snap"],
                               edge_visits_per_step_per_gpu=wb.n_edge_visits, node_visits_per_step_per_gpu=wb.n_node_visits,
                               distinct_snapshot_edges_per_step=getattr(wb, "n_edges_distinct", None),
                               distinct_snapshot_nodes_per_step=getattr(wb, "n_nodes_distinct", None), targets=targets,
                               parallelism=("snapshot-visits/%d+allgather(node states)+grad-allreduce" % world) if sharded
                               else ("dp%d(windows)+grad-allreduce" % world), host_prepare_s=prepare_s,
                               launch=(("3 hip graphs + eager collectives" if sharded_step.graphs is not None else "eager") if sharded
                                       else ("hip-graph replay" if graph is not None else "eager")),
                               rccl_ranks=(dist.get_world_size() if dist is not None else 0),
                               train_loop=loop,
                               mfma=("large fp32 products on the 16-bit matrix pipes with fp32-equivalent accuracy: three f16 MFMA products of the scaled "
                                     "2-way operand split (split_f16.hpp) in the window-chain kernels, the GRU weight gradients, the input-gate and "
                                     "d_x GEMMs; six bf16 products of the exact 3-way split (gemm_bx.hpp) in the self-loop products and the loop-weight "
                                     "gradient; small products (< 16384 rows) on the fp32 MFMA pipe" if MFMA_MODE == "f16x2" else
                                     "every large fp32 product (panel GEMMs, weight gradients, the window-chain kernels' W_hh products) on the bf16 "
                                     "matrix pipe as six products of an exact 3-way operand split (fp32-equivalent accuracy, gemm_bx.hpp); "
                                     "small products (< 16384 rows) on the fp32 MFMA pipe" if MFMA_MODE == "bf16x3"
                                     else "fp32 MFMA everywhere (TEMP_OPT_MFMA_BF16X3 = 0)"),
                               fp32_mfma_ms_per_step=fp32_ms, ms_per_step_400_replays=long_ms),
                   roofline=roof, cpu_baseline=cpu, north_star_sharded=ns_result, extra=extra)
        # the end-to-end figures a reader of the line's TAIL must see (the driver keeps the parsed head and the last characters):
        # value_distinct and the training loop (a NEW batch every step) also as the LAST keys of `extra`
        if not sharded:
            ex = dict(out["extra"] or {})
            ex["value_distinct"] = out["value_distinct"]
            ex["train_loop"] = loop
            out["extra"] = ex
        # RCCL writes a version banner through C stdio (block-buffered when stdout is a pipe): push it out first so
        # that the JSON line is the LAST (and, with claim_stdout, the only) line of stdout
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
