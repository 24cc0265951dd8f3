"""world_size-2 and -4 gloo tests (CPU, test-only kernel backend) of both multi-GPU decompositions:
windows -> ranks with a gradient all-reduce, and snapshot visits -> ranks with the all-gather of
per-snapshot node states before the recurrent chain.  Both must reproduce the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden_util import load


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(name, batched=True):
    from temp_amd import backend as TB
    from tests.cpu_backend import CpuTestBackend
    from tests.window_cases import build_window_model, window_inputs
    TB.set_backend(CpuTestBackend())
    z = load(name)
    m = build_window_model(z, torch.device("cpu"), batched)
    edge_ids, _ = window_inputs(z)
    t_list = sorted([int(t) for t in z["t_list"]], reverse=True)
    return m, z, edge_ids, t_list


def _reference(name):
    m, z, edge_ids, t_list = _build(name)
    per_graph, *_ = m.encode(torch.tensor(t_list), int(z["L"]), True, edge_ids)
    loss = sum((e * (i + 1)).sum() for i, e in enumerate(per_graph))
    loss.backward()
    return [e.detach() for e in per_graph], {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}


def _worker(rank, world, port, name, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from temp_amd.dist import SnapshotShardedEncoder, allreduce_gradients
        m, z, edge_ids, t_list = _build(name)
        L = int(z["L"])
        if mode == "snapshots":
            enc = SnapshotShardedEncoder(m)
            sb = enc.prepare(torch.tensor(t_list), L, True, edge_ids)
            # one group per GRU, disjoint x rows: the shape on which the chain backward writes its gate gradients once and
            # every weight gradient is one launch (the shape of the one-GPU program; gru_chain.py backward)
            groups = sb.program.groups
            assert len(groups) == len({g["rnn"] for g in groups}) <= 2
            assert all(a["x1"] <= b["x0"] for a, b in zip(groups, groups[1:]))
            out = enc.run(sb)
            pieces = list(out.split(sb.target_sizes)) if sb.target_sizes else []
            wins = sb.target_windows
        else:                                   # windows -> ranks: each rank encodes its own windows
            wins = [b for b in range(len(t_list)) if b % world == rank]
            sub_t = [t_list[b] for b in wins]
            pieces = []                         # a rank that owns no window of a short batch does no encoder work at all
            if wins:
                pieces, *_ = m.encode(torch.tensor(sub_t), L, True, [edge_ids[b] for b in wins])
        loss = sum((e * (b + 1)).sum() for b, e in zip(wins, pieces))
        if mode == "snapshots" and not pieces:  # a rank without a window of the recurrence still owns RGCN shards: its (empty)
            loss = out.sum()                    # output keeps it in the backward collectives (reduce-scatter of the node states)
        if isinstance(loss, torch.Tensor):
            loss.backward()
        # ranks without work (or without a gradient for some parameter) still join the SAME collective: the bucket zero-fills
        allreduce_gradients(list(m.parameters()), world, average=False)
        q.put((rank, wins, [e.detach().numpy() for e in pieces], {k: v.grad.numpy() for k, v in m.named_parameters() if v.grad is not None}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,name,world", [("windows", "G10_bi_grrgcn_rol", 2), ("snapshots", "G10_bi_grrgcn_rol", 2),
                                             ("snapshots", "G10_uni_grrgcn_rol", 2), ("windows", "G10_uni_grrgcn", 2),
                                             # 3 windows on 4 ranks: an idle rank (windows mode) / uneven shards and a rank
                                             # without a window of the recurrence (snapshots mode)
                                             ("windows", "G10_uni_grrgcn_rol", 4), ("snapshots", "G10_uni_grrgcn_rol", 4),
                                             ("snapshots", "G10_bi_grrgcn_rol", 4)])
def test_ranks_match_single_process(mode, name, world):
    ref_out, ref_grads = _reference(name)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    import queue as _queue
    import time as _time
    t_end = _time.time() + 600
    while len(results) < world:
        try:
            results.append(q.get(timeout=2))
        except _queue.Empty:
            assert _time.time() < t_end, "timed out"
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), "a rank died: %s" % [p.exitcode for p in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = set()
    for rank, wins, pieces, grads in results:
        for b, e in zip(wins, pieces):
            np.testing.assert_allclose(e, ref_out[b].numpy(), rtol=2e-5, atol=2e-6)
            seen.add(b)
        # a parameter no rank produced a gradient for keeps .grad = None, as in the single-process run (presence mask of GradBucket)
        assert set(grads) == set(ref_grads), (sorted(set(grads) ^ set(ref_grads)))
        for k, g in grads.items():
            np.testing.assert_allclose(g, ref_grads[k].numpy(), rtol=2e-4, atol=3e-6, err_msg=k)
    assert seen == set(range(len(ref_out)))


def test_split_visits_by_edges():
    from temp_amd.dist import split_visits_by_edges
    b = split_visits_by_edges([10, 10, 10, 10, 40, 10, 10], 4)
    assert b[0] == 0 and b[-1] == 7 and all(x <= y for x, y in zip(b, b[1:]))
    assert split_visits_by_edges([], 3) == [0, 0, 0, 0]
    assert split_visits_by_edges([5], 2)[-1] == 1


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from temp_amd.dist import allreduce_gradients
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2))]
        g = [torch.full((3, 4), 1.0 + rank), torch.full((5,), 10.0 * (rank + 1)), torch.full((2, 2), -1.0 - rank)]
        ps[0].grad = g[0].clone()
        ps[1].grad = g[1].clone() if rank == 0 else None          # rank 1 has no gradient for this parameter: zeros join the sum
        ps[2].grad = g[2].clone()
        allreduce_gradients(ps, world, average=True)
        first = [p.grad.clone() for p in ps]
        # the gradients now live in the bucket: an in-place accumulation (no zero_grad between two backward passes) must be seen
        # by the next call, and a parameter whose .grad was replaced must be picked up again
        ps[0].grad += 1.0 + rank
        ps[2].grad = torch.full((2, 2), 7.0 * (rank + 1))
        allreduce_gradients(ps, world, average=False)
        second = [p.grad.clone() for p in ps]
        # explicit sources (a replayed graph's capture tensors) win over .grad
        src = [torch.full((3, 4), 2.0), None, torch.full((2, 2), 3.0 * (rank + 1))]
        allreduce_gradients(ps, world, average=False, grads=src)
        third = [None if p.grad is None else p.grad.clone() for p in ps]
        q.put((rank, [t.numpy() for t in first], [t.numpy() for t in second], [None if t is None else t.numpy() for t in third]))
    finally:
        dist.destroy_process_group()


def test_gradient_bucket_semantics():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, first, second, third in results:
        np.testing.assert_allclose(first[0], np.full((3, 4), 1.5))            # (1 + 2) / 2
        np.testing.assert_allclose(first[1], np.full((5,), 5.0))              # (10 + 0) / 2
        np.testing.assert_allclose(first[2], np.full((2, 2), -1.5))
        np.testing.assert_allclose(second[0], np.full((3, 4), 1.5 + 1 + 1.5 + 2))   # both ranks' (average + own increment), summed
        np.testing.assert_allclose(second[1], np.full((5,), 10.0))            # the averaged span of both ranks, summed
        np.testing.assert_allclose(second[2], np.full((2, 2), 21.0))
        np.testing.assert_allclose(third[0], np.full((3, 4), 4.0))
        assert third[1] is None      # NO rank had a gradient for it: .grad stays None as in single-process training (Adam skips it)
        np.testing.assert_allclose(third[2], np.full((2, 2), 9.0))


def _step_worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from temp_amd.dist import ShardedStep, SnapshotShardedEncoder
        m, z, edge_ids, t_list = _build(name)
        enc = SnapshotShardedEncoder(m)
        sb = enc.prepare(torch.tensor(t_list), int(z["L"]), True, edge_ids)
        st = ShardedStep(enc, sb, list(m.parameters()), graphs=False, average=False)
        out = st.step()
        pieces = list(out.detach().split(sb.target_sizes)) if sb.target_sizes else []
        q.put((rank, sb.target_windows, [e.numpy() for e in pieces], {k: v.grad.numpy() for k, v in m.named_parameters() if v.grad is not None}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("G10_bi_grrgcn_rol", 2), ("G10_uni_grrgcn_rol", 4)])
def test_sharded_step_parts_match_single_process(name, world):
    """ShardedStep (the three compute parts around the two exchanges that bench.py replays as HIP graphs on the GPU; eager here)
    with upstream gradient = ones: target embeddings and all-reduced gradients equal the single-process step's."""
    m, z, edge_ids, t_list = _build(name)
    per_graph, *_ = m.encode(torch.tensor(t_list), int(z["L"]), True, edge_ids)
    sum(e.sum() for e in per_graph).backward()
    ref_out = [e.detach().numpy() for e in per_graph]
    ref_grads = {k: v.grad.numpy() for k, v in m.named_parameters() if v.grad is not None}
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = set()
    for rank, wins, pieces, grads in results:
        for b, e in zip(wins, pieces):
            np.testing.assert_allclose(e, ref_out[b], rtol=2e-5, atol=2e-6)
            seen.add(b)
        assert set(grads) == set(ref_grads)
        for k, g in grads.items():
            np.testing.assert_allclose(g, ref_grads[k], rtol=2e-4, atol=3e-6, err_msg=k)
    assert seen == set(range(len(ref_out)))
