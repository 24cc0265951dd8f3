"""Window-level parity cases shared by CPU (test backend) and GPU (HIP) runs: drive
DynamicRGCN / BiDynamicRGCN / StaticRGCN.forward on the committed ICEWS14 slice with the reference's
recorded random draws and compare loss + gradients with the golden vectors (G10, G12)."""
import argparse

import numpy as np
import torch

from oracle import temp_oracle as O
from temp_amd.bi_dynamic_rgcn import BiDynamicRGCN
from temp_amd.dataset import build_interpolation_snapshots
from temp_amd.dynamic_rgcn import DynamicRGCN
from temp_amd.self_attention_rgcn import BiSelfAttentionRGCN, SelfAttentionRGCN
from temp_amd.static_rgcn import StaticRGCN
from tests.golden_util import T, assert_close, checksum, load

_SLICE = {}


def slice_snapshots():
    if not _SLICE:
        z = load("icews14_slice")
        tr, va, te = build_interpolation_snapshots(z["train"], z["valid"], z["test"], z["times"])
        _SLICE.update(num_e=int(z["num_ents"]), num_r=int(z["num_rels"]), times=[int(t) for t in z["times"]], tr=tr, va=va, te=te)
    return _SLICE


def make_args(**over):
    d = dict(n_bases=16, dropout=0.0, inv_temperature=0.1, learnable_lambda=False, impute=False, post_aggregation=False,
             post_ensemble=False, num_layers=1, type1=False, rec_only_last_layer=False, use_time_embedding=False,
             module='GRRGCN', embed_size=32, hidden_size=32, num_pos_facts=3000, negative_rate=20, score_function='complex',
             train_seq_len=8, test_seq_len=8, use_cuda=False, debug=False, edge_dropout=False, random_dropout=False,
             use_embed_for_non_active=False, lr=1e-3, seed=0, batch_size=4)
    d.update(over)
    return argparse.Namespace(**d)


def state_dict_from_oracle(model):
    """oracle parameter dict -> the REFERENCE's state_dict key names."""
    sd = {'ent_embeds': model['ent_embeds'], 'rel_embeds': model['rel_embeds']}
    for ln, d in model['ent_encoder'].items():
        p = 'ent_encoder.%s.' % ln
        for k in ('weight', 'loop_weight', 'time_embed', 'time_weight', 'time_weight_forward', 'time_weight_backward', 'h_bias'):
            if d.get(k) is not None:
                sd[p + k] = d[k]
        for k in ('q_linear', 'k_linear', 'v_linear'):
            if k in d:
                sd[p + k + '.weight'] = d[k]
        if 'exponential_decay' in d:
            sd[p + 'exponential_decay.weight'], sd[p + 'exponential_decay.bias'] = d['exponential_decay']
        for name in ('rnn', 'forward_rnn', 'backward_rnn'):
            if name in d:
                for li, q in enumerate(d[name]):
                    for a, b in (('w_ih', 'weight_ih'), ('w_hh', 'weight_hh'), ('b_ih', 'bias_ih'), ('b_hh', 'bias_hh')):
                        sd['%s%s.%s_l%d' % (p, name, b, li)] = q[a]
    return sd


def build_window_model(z, device, batched=True, chain=True, stack=True):
    s = slice_snapshots()
    module, rec_only, D, B, L = str(z["module"]), bool(z["rec_only"]), int(z["D"]), int(z["B"]), int(z["L"])
    cfg = dict(module=module, n_bases=B, inv_temperature=0.1, rec_only_last_layer=rec_only, use_time_embedding=bool(z["te"]))
    model = O.init_model(cfg, s["num_e"], s["num_r"], len(s["times"]), D, seed=int(z["seed"]))
    if "rel_scale" in z.files:              # G13: relation embeddings scaled so that the scores spread (oracle/gen_golden.py:G13_REL_SCALE)
        model["rel_embeds"] = model["rel_embeds"] * float(z["rel_scale"])
    assert abs(checksum(model) - float(z["param_checksum"])) < 1e-6 * max(1.0, float(z["param_checksum"]) * 1e-6)
    args = make_args(module=module, rec_only_last_layer=rec_only, embed_size=D, hidden_size=D, n_bases=B, train_seq_len=L,
                     test_seq_len=L, negative_rate=int(z["neg"]), use_time_embedding=bool(z["te"]))
    cls = BiDynamicRGCN if module.startswith("Bi") else DynamicRGCN
    m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"])
    m.load_state_dict(state_dict_from_oracle(model), strict=True)
    m.use_batched_path = batched
    m.use_gru_chain = chain
    m.use_rec_stack = stack
    return m.to(device)


def window_inputs(z):
    n = int(z["n_choices"])
    edge_ids = [z["choice_%d" % i] for i in range(n)]
    samples = [(T(z["trip_%d" % i]).long(), T(z["negtail_%d" % i]).long(), T(z["neghead_%d" % i]).long()) for i in range(n)]
    return edge_ids, samples


def check_window(name, device, batched=True, stack=True):
    z = load(name)
    m = build_window_model(z, device, batched, stack=stack)
    edge_ids, samples = window_inputs(z)
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    loss = m(t_list, target_edge_ids=edge_ids, samples=samples)
    want = float(z["loss"])
    assert abs(loss.item() - want) < 3e-5 * abs(want), (name, loss.item(), want)
    loss.backward()
    eg = m.ent_embeds.grad
    rows = T(z["d_ent_nz_rows"]).long().to(device)
    assert_close(eg[rows], z["d_ent_nz_vals"], 1e-4, 3e-6, name + " d_ent")
    if "d_ent_sub" not in z.files:          # full list of non-zero rows: every other row must be exactly untouched
        mask = torch.ones(eg.shape[0], dtype=torch.bool, device=device)
        mask[rows] = False
        assert float(eg[mask].abs().max()) < 1e-7
    assert_close(m.rel_embeds.grad, z["d_rel"], 1e-4, 3e-6, name + " d_rel")
    checked = 0
    for k, v in m.named_parameters():
        gk = "gabs_" + k
        if gk in z.files and v.grad is not None:
            want = float(z[gk])
            got = v.grad.double().abs().sum().item()
            assert abs(got - want) < 3e-4 * max(want, 1e-3), (name, k, got, want)
            checked += 1
    assert checked >= 5
    return m


def check_batched_equals_generic(name, device):
    """The batched path (one RGCN launch per layer over all visits + GRU chain through prev_idx)
    must reproduce the reference-granular path."""
    z = load(name)
    edge_ids, samples = window_inputs(z)
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    outs = []
    for batched, chain in ((False, False), (True, False), (True, True)):
        m = build_window_model(z, device, batched, chain)
        assert m._can_batch() == batched and m._can_chain() == chain
        wb = m.prepare(t_list, int(z["L"]), True, edge_ids)
        assert (wb.program is not None) == (batched and chain)
        per_graph, *_ = m.encode(t_list, int(z["L"]), True, edge_ids)
        s = sum((e * (i + 1)).sum() for i, e in enumerate(per_graph))
        s.backward()
        outs.append((per_graph, m.ent_embeds.grad.clone(), m.ent_encoder.layer_2.weight.grad.clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0][0], other[0]):
            assert_close(a, b, 1e-5, 2e-6, name + " batched vs generic")
        assert_close(outs[0][1], other[1], 5e-5, 3e-6, name + " d_ent batched vs generic")
        assert_close(outs[0][2], other[2], 5e-5, 3e-6, name + " d_weight batched vs generic")


def check_stack_equals_generic(name, device, type1=False, width=None):
    """Both layers recurrent (the reference's default flags): the one-node position loop (temp_amd/rec_stack.py) must
    reproduce the reference-granular loop of RRGCN.forward calls -- loss, embedding / relation gradients and the gradient of
    every encoder parameter -- on the golden's windows, edge subsamples and negative samples."""
    z = load(name)
    assert not bool(z["rec_only"])
    edge_ids, samples = window_inputs(z)
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    res = []
    for stack in (False, True):
        if type1 or width:                         # (no golden for the type-1 cell / this width at window level: same seed, both paths)
            s = slice_snapshots()
            D, B, L = int(width or z["D"]), (int(width) // 2 if width else int(z["B"])), int(z["L"])
            args = make_args(module=str(z["module"]), rec_only_last_layer=False, embed_size=D, hidden_size=D, n_bases=B, train_seq_len=L,
                             test_seq_len=L, negative_rate=int(z["neg"]), type1=type1)
            torch.manual_seed(11)
            m = DynamicRGCN(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(device)
            for k, p in m.named_parameters():      # the cell draws its weights from N(0, 1) like the reference's: two stacked recurrent
                if "rnn" in k and type1:           # layers over 8 positions amplify fp32 summation-order noise to 3e-4 of the gradients;
                    p.data.mul_(0.2)               # at a fifth of the scale the two paths agree to 1e-7 (same arithmetic, other order)
        else:
            m = build_window_model(z, device, True)
        m.use_rec_stack = stack
        wb = m.prepare(t_list, int(z["L"]), True, edge_ids)
        assert bool(wb.stack) == stack and not wb.batched and (wb.program is not None) == stack
        loss = m.run_loss(wb, samples)
        loss.backward()
        res.append((loss.detach().cpu(), {k: v.grad.detach().cpu().clone() for k, v in m.named_parameters() if v.grad is not None}))
    (l0, g0), (l1, g1) = res
    assert abs(l0.item() - l1.item()) < 2e-5 * abs(l0.item()), (l0.item(), l1.item())
    assert set(g0) == set(g1) and len(g0) >= 10
    for k in g0:
        assert_close(g1[k], g0[k], 1e-4, 3e-6 * max(1.0, float(g0[k].abs().max())), name + " stack vs generic: d_" + k)


def check_static(device):
    z = load("G12_static_rgcn")
    s = slice_snapshots()
    D, B, seed = int(z["D"]), int(z["B"]), int(z["seed"])
    cfg = dict(module="SRGCN", n_bases=B, inv_temperature=0.1, rec_only_last_layer=False, use_time_embedding=False)
    model = O.init_model(cfg, s["num_e"], s["num_r"], len(s["times"]), D, seed=seed, bias=True)
    rng = np.random.default_rng(seed + 1)
    for ln in ("layer_1", "layer_2"):
        model["ent_encoder"][ln]["h_bias"] = T(rng.uniform(-0.3, 0.3, D).astype(np.float32))
    args = make_args(module="SRGCN", embed_size=D, hidden_size=D, n_bases=B)
    m = StaticRGCN(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"])
    m.load_state_dict(state_dict_from_oracle(model), strict=True)
    m.to(device)
    tl = [int(t) for t in z["t_list"]]
    with torch.no_grad():
        embeds = m.get_per_graph_ent_embeds(tl, [s["tr"][t] for t in tl], val=True)
        iso = m.ent_encoder.forward_isolated(m.ent_embeds[:200], tl[0])
    for i, e in enumerate(embeds):
        assert_close(e, z["emb_%d" % i], 1e-5, 2e-6, "G12 emb %d" % i)
    assert_close(iso, z["iso"], 1e-5, 2e-6, "G12 iso")
    # training step with the reference's recorded draws: golden loss + gradients (baselines/StaticRGCN.py:36-46)
    edge_ids, samples = window_inputs(z)
    loss = m(torch.tensor(tl), target_edge_ids=edge_ids, samples=samples)
    want = float(z["loss"])
    assert abs(loss.item() - want) < 3e-5 * abs(want), (loss.item(), want)
    loss.backward()
    eg = m.ent_embeds.grad
    rows = T(z["d_ent_nz_rows"]).long().to(device)
    assert_close(eg[rows], z["d_ent_nz_vals"], 1e-4, 3e-6, "G12 d_ent")
    mask = torch.ones(eg.shape[0], dtype=torch.bool, device=device)
    mask[rows] = False
    assert float(eg[mask].abs().max()) < 1e-7
    assert_close(m.rel_embeds.grad, z["d_rel"], 1e-4, 3e-6, "G12 d_rel")
    for ln in ("layer_1", "layer_2"):
        assert_close(getattr(m.ent_encoder, ln).h_bias.grad, z["d_bias_" + ln], 1e-4, 3e-6, "G12 d_bias " + ln)
    checked = 0
    for k, v in m.named_parameters():
        gk = "gabs_" + k
        if gk in z.files and v.grad is not None:
            w = float(z[gk])
            assert abs(v.grad.double().abs().sum().item() - w) < 3e-4 * max(w, 1e-3), (k, w)
            checked += 1
    assert checked >= 8, checked
    m.zero_grad()
    loss = m(torch.tensor(tl))            # and end-to-end with its own sampler (fused loss path): finite, differentiable
    loss.backward()
    assert torch.isfinite(loss) and m.ent_embeds.grad.abs().sum() > 0
    ranks, ev_loss = m.evaluate(torch.tensor(tl))
    assert ranks.numel() > 0 and int(ranks.min()) >= 1 and int(ranks.max()) <= s["num_e"] and np.isfinite(ev_loss)


def check_evaluate(name, device):
    """evaluate(): filtered ranks and loss against the reference's own evaluate() on the ICEWS14 slice (G13).

    Integer parity: the fixture records, for every ranked row, how many unfiltered competitors have a sigmoid score within
    `band` (1.5e-6, a few fp32 ulps at 0.5) of the target's -- pairs that two valid fp32 evaluation orders may swap (and
    that the reference itself resolves by whatever order torch.sort returns).  Rows with no such competitor (about 90 %)
    must match the reference's rank EXACTLY; any other row may differ by at most its number of in-band competitors."""
    z = load(name)
    m = build_window_model(z, device)
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    for split, val in (("val", True), ("test", False)):
        ranks, loss = m.evaluate(t_list, val=val)
        want = T(z["ranks_" + split]).long()
        nclose = T(z["nclose_" + split]).long()
        assert ranks.shape == want.shape
        got = ranks.cpu()
        safe = nclose == 0
        assert safe.float().mean().item() > 0.85, (name, split)
        assert torch.equal(got[safe], want[safe]), (name, split, int((got[safe] != want[safe]).sum()))
        assert bool(((got - want).abs() <= nclose).all()), (name, split)
        assert abs(loss - float(z["loss_" + split])) < 2e-5 * max(1.0, abs(float(z["loss_" + split])))


def build_sa_model(z, device):
    s = slice_snapshots()
    module, rec_only, D, B, L, learn = str(z["module"]), bool(z["rec_only"]), int(z["D"]), int(z["B"]), int(z["L"]), bool(z["learn"])
    cfg = dict(module=module, n_bases=B, inv_temperature=0.1, rec_only_last_layer=rec_only, use_time_embedding=True, learnable_lambda=learn)
    model = O.init_model(cfg, s["num_e"], s["num_r"], len(s["times"]), D, seed=int(z["seed"]))
    if learn:
        for ln in ("layer_1", "layer_2"):
            model["ent_encoder"][ln]["exponential_decay"] = (torch.full((1, 1), 0.25), torch.full((1,), -0.1))
    assert abs(checksum(model) - float(z["param_checksum"])) < 1e-6
    args = make_args(module=module, rec_only_last_layer=rec_only, embed_size=D, hidden_size=D, n_bases=B, train_seq_len=L,
                     test_seq_len=L, negative_rate=int(z["neg"]), learnable_lambda=learn, EMA=False)
    cls = BiSelfAttentionRGCN if module.startswith("Bi") else SelfAttentionRGCN
    m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"])
    sd = state_dict_from_oracle(model)
    if learn and rec_only:
        sd.pop("ent_encoder.layer_1.exponential_decay.weight", None)
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("exponential_decay" in k for k in missing.missing_keys), missing
    return m.to(device), model, cfg


def check_sa_window(name, device):
    """SelfAttentionRGCN / BiSelfAttentionRGCN.forward: loss + gradients against the reference (G14)."""
    z = load(name)
    m, _, _ = build_sa_model(z, device)
    edge_ids, samples = window_inputs(z)
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    loss = m(t_list, target_edge_ids=edge_ids, samples=samples)
    want = float(z["loss"])
    assert abs(loss.item() - want) < 3e-5 * abs(want), (name, loss.item(), want)
    loss.backward()
    eg = m.ent_embeds.grad
    rows = T(z["d_ent_nz_rows"]).long().to(device)
    assert_close(eg[rows], z["d_ent_nz_vals"], 1e-4, 3e-6, name + " d_ent")
    assert_close(m.rel_embeds.grad, z["d_rel"], 1e-4, 3e-6, name + " d_rel")
    checked = 0
    for k, v in m.named_parameters():
        gk = "gabs_" + k
        if gk in z.files and v.grad is not None:
            want = float(z[gk])
            got = v.grad.double().abs().sum().item()
            assert abs(got - want) < 3e-4 * max(want, 1e-3), (name, k, got, want)
            checked += 1
    assert checked >= 9, checked
    return m


def check_sa_dense_api(device, seed=5):
    """SARGCNLayer.calc_result / SARGCN.forward_isolated through the reference's DENSE signature
    (history tensor + additive mask) against the oracle restatement."""
    z = load("G14_sa_uni")
    m, model, cfg = build_sa_model(z, device)
    g = torch.Generator().manual_seed(seed)
    n, Th, D = 37, 4, int(z["D"])
    cur = torch.randn(n, D, generator=g)
    live = torch.rand(n, Th, generator=g) < 0.4
    prev = torch.randn(n, Th, D, generator=g) * live.unsqueeze(-1)
    mask = torch.cat([torch.where(live, 0.0, -10e9), torch.zeros(n, 1)], dim=1)
    td = torch.arange(Th, -1, -1, dtype=torch.float32)
    want = O.sa_attention(model["ent_encoder"]["layer_2"], cfg, cur, prev, td, mask)
    got = m.ent_encoder.layer_2.calc_result(cur.to(device), prev.to(device), td.to(device), mask.to(device))
    assert_close(got, want, 1e-5, 2e-6, "calc_result dense")
    e = torch.randn(n, D, generator=g)
    want = O.sargcn_isolated(model["ent_encoder"], cfg, e, prev, prev * 0.5, td, mask, 3)
    got = m.ent_encoder.forward_isolated(e.to(device), prev.to(device), (prev * 0.5).to(device), td.to(device), mask.to(device), 3)
    assert_close(got, want, 1e-5, 2e-6, "forward_isolated dense")


def check_sa_evaluate(name, device):
    """evaluate() of the attention models: ranks in range, and the encoder outputs it ranks with equal
    the oracle's (full target graphs, no sampling)."""
    z = load(name)
    m, model, cfg = build_sa_model(z, device)
    s = slice_snapshots()
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    ranks, loss = m.evaluate(t_list)
    assert ranks.numel() > 0 and int(ranks.min()) >= 1 and int(ranks.max()) <= s["num_e"] and np.isfinite(loss)
    with torch.no_grad():
        per_graph, wb, tables = m.encode(t_list, int(z["L"]), train=False)
        all_list = m.all_embeds_batched(wb, torch.cat(per_graph, dim=0), tables)
    from tests.test_oracle_golden import slice_graphs
    _, _, times, gd = slice_graphs()
    tl = sorted([int(t) for t in z["t_list"]], reverse=True)
    targets = [gd["train"][t] for t in tl]
    with torch.no_grad():
        want, tt, hist, mask, td = O.sa_encode(model, cfg, gd["train"], tl, times, int(z["L"]), targets, bi=cfg["module"].startswith("Bi"))
        for i, (a, b) in enumerate(zip(per_graph, want)):
            assert_close(a, b, 1e-5, 2e-6, name + " eval encode")
            assert_close(all_list[i], O.sa_all_embeds(model, cfg, gd["train"], i, tt[i], b, hist, mask, td), 1e-5, 2e-6, name + " all embeds")


# ---------------------------------------------------------------------------------------------------------------------
# config 3: post-ensemble / impute window models (G15)
# ---------------------------------------------------------------------------------------------------------------------
IMPUTE_GATES = {"impute_weight": (0.3, -0.1), "impute_weight_forward": (0.25, -0.05), "impute_weight_backward": (0.4, 0.1)}


def build_post_model(z, device, cls, batched=True, **flags):
    s = slice_snapshots()
    module, rec_only, D, B, L = str(z["module"]), bool(z["rec_only"]), int(z["D"]), int(z["B"]), int(z["L"])
    cfg = dict(module=module, n_bases=B, inv_temperature=0.1, rec_only_last_layer=rec_only, use_time_embedding=False)
    model = O.init_model(cfg, s["num_e"], s["num_r"], len(s["times"]), D, seed=int(z["seed"]))
    assert abs(checksum(model) - float(z["param_checksum"])) < 1e-6
    args = make_args(module=module, rec_only_last_layer=rec_only, embed_size=D, hidden_size=D, n_bases=B, train_seq_len=L,
                     test_seq_len=L, negative_rate=int(z["neg"]), **flags)
    m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"])
    missing = m.load_state_dict(state_dict_from_oracle(model), strict=False)
    assert not missing.unexpected_keys and all(("impute_weight" in k or "_linear." in k) for k in missing.missing_keys), missing
    with torch.no_grad():
        for nm, (w, b) in IMPUTE_GATES.items():
            if hasattr(m.ent_encoder, nm):
                getattr(m.ent_encoder, nm).weight.fill_(w)
                getattr(m.ent_encoder, nm).bias.fill_(b)
    m.use_batched_path = batched
    return m.to(device)


def check_post_bi(device, batched=True):
    """Config 3's model at window level (G15_post_bi, recorded from the reference's ImputeBiDynamicRGCN.pre_forward /
    get_final_graph_embeds + BiRRGCN.forward_post_ensemble_isolated): (local, temporal) target embeddings, local history
    streams, all-entity (local, temporal) matrices, gradients of their seeded weighted sum."""
    from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN
    z = load("G15_post_bi")
    m = build_post_model(z, device, PostEnsembleBiDynamicRGCN, batched, post_ensemble=True)
    assert m._can_batch() == batched
    edge_ids = [z["choice_%d" % i] for i in range(int(z["n_choices"]))]
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    L = int(z["L"])
    locs, recs, wb, hist = m.encode_post(t_list, L, True, edge_ids)
    assert wb.batched == batched and (wb.program is not None) == batched
    rows_sel = T(z["all_rows"]).long().to(device)
    gen = torch.Generator().manual_seed(int(z["seed"]))
    total = 0
    N = m.num_ents
    for i, g in enumerate(wb.graphs):
        assert_close(locs[i], z["loc_%d" % i], 1e-5, 2e-6, "G15 loc %d" % i)
        assert_close(recs[i], z["rec_%d" % i], 1e-5, 2e-6, "G15 rec %d" % i)
        # local history streams: the rows of the last executed position, scattered to entity ids
        for nm, plan, loc in (("f_loc", wb.plan[0], wb.hist_loc[0]), ("b_loc", wb.plan[1], wb.hist_loc[1])):
            row_of, _ = plan.final_all(i, L - 1)
            have = np.nonzero(row_of >= 0)[0]
            want_rows = z["%s_%d_rows" % (nm, i)]
            vals = loc[torch.from_numpy(row_of[have]).long().to(device)] if have.size else torch.zeros(0, locs[i].shape[1])
            nz = (vals.detach().abs().sum(1) > 0).cpu().numpy() if have.size else np.zeros(0, bool)     # a ReLU row can be all zero
            assert np.array_equal(have[nz], want_rows), (nm, i)
            assert_close(vals[torch.from_numpy(nz).to(vals.device)], z["%s_%d_vals" % (nm, i)], 1e-5, 2e-6, "G15 %s %d" % (nm, i))
        t = wb.rows[i][-1]
        full = m.graph_dict_train[t]
        a_loc, a_rec = m.get_all_embeds_Gt(locs[i], recs[i], g, t, wb.plan, i, hist, wb.hist_loc)
        # the fixture holds the isolated pass BEFORE the active rows are written over it: compare the inactive rows, and the
        # active ones against the encoder outputs they were overwritten with
        act = torch.zeros(N, dtype=torch.bool, device=device)
        act[torch.from_numpy(g.gids).to(device)] = True
        keep = ~act[rows_sel]
        assert_close(a_loc[rows_sel][keep], T(z["all_loc_%d" % i]).to(device)[keep], 1e-5, 2e-6, "G15 all loc")
        assert_close(a_rec[rows_sel][keep], T(z["all_rec_%d" % i]).to(device)[keep], 1e-5, 2e-6, "G15 all rec")
        assert_close(a_loc[torch.from_numpy(g.gids).to(device)], locs[i], 0, 0, "active rows")
        iso_loc, iso_rec = m.get_all_embeds_Gt(locs[i][:0], recs[i][:0], type(g)(0, [], [], [], []), t, wb.plan, i, hist, wb.hist_loc)
        for x, r in ((locs[i], None), (recs[i], None), (iso_loc[rows_sel], None), (iso_rec[rows_sel], None)):
            total = total + (x * torch.randn(x.shape, generator=gen).to(device)).sum()
    want = float(z["total"])
    assert abs(total.item() - want) < 3e-5 * max(1.0, abs(want)), (total.item(), want)
    total.backward()
    eg = m.ent_embeds.grad
    assert_close(eg[T(z["d_ent_nz_rows"]).long().to(device)], z["d_ent_nz_vals"], 1e-4, 3e-6, "G15 d_ent")
    checked = 0
    for k, v in m.named_parameters():
        gk = "gabs_" + k
        if gk in z.files and v.grad is not None:
            w = float(z[gk])
            assert abs(v.grad.double().abs().sum().item() - w) < 3e-4 * max(w, 1e-3), (k, w)
            checked += 1
    assert checked >= 9, checked
    return m


def check_impute_window(name, device, batched=True):
    """ImputeDynamicRGCN / ImputeBiDynamicRGCN.forward (--impute): loss + gradients against the reference (G15_impute_*)."""
    from temp_amd.post_dynamic_rgcn import ImputeBiDynamicRGCN, ImputeDynamicRGCN
    z = load(name)
    bi = str(z["module"]).startswith("Bi")
    m = build_post_model(z, device, ImputeBiDynamicRGCN if bi else ImputeDynamicRGCN, batched, impute=True)
    edge_ids, samples = window_inputs(z)
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    loss = m(t_list, target_edge_ids=edge_ids, samples=samples)
    want = float(z["loss"])
    assert abs(loss.item() - want) < 3e-5 * abs(want), (name, loss.item(), want)
    loss.backward()
    eg = m.ent_embeds.grad
    assert_close(eg[T(z["d_ent_nz_rows"]).long().to(device)], z["d_ent_nz_vals"], 1e-4, 3e-6, name + " d_ent")
    assert_close(m.rel_embeds.grad, z["d_rel"], 1e-4, 3e-6, name + " d_rel")
    checked = 0
    for k, v in m.named_parameters():
        gk = "gabs_" + k
        if gk in z.files and v.grad is not None:
            w = float(z[gk])
            assert abs(v.grad.double().abs().sum().item() - w) < 3e-4 * max(w, 1e-3), (name, k, v.grad.double().abs().sum().item(), w)
            checked += 1
    assert checked >= 7, checked
    return m


def check_impute_evaluate(name, device, batched=True):
    """evaluate() of the impute models against the reference's own ranks and loss (G16): exact wherever no competitor sits in the
    fp32 tie band of the target, within the band population elsewhere (as check_evaluate)."""
    from temp_amd.post_dynamic_rgcn import ImputeBiDynamicRGCN, ImputeDynamicRGCN
    z = load(name)
    bi = str(z["module"]).startswith("Bi")
    m = build_post_model(z, device, ImputeBiDynamicRGCN if bi else ImputeDynamicRGCN, batched, impute=True)
    with torch.no_grad():
        m.rel_embeds.mul_(float(z["rel_scale"]))
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    for split, val in (("val", True), ("test", False)):
        ranks, loss = m.evaluate(t_list, val=val)
        want = T(z["ranks_" + split]).long()
        nclose = T(z["nclose_" + split]).long()
        assert ranks.shape == want.shape
        got = ranks.cpu()
        safe = nclose == 0
        assert safe.float().mean().item() > 0.75, (name, split)
        assert torch.equal(got[safe], want[safe]), (name, split, int((got[safe] != want[safe]).sum()))
        assert bool(((got - want).abs() <= nclose).all()), (name, split)
        assert abs(loss - float(z["loss_" + split])) < 2e-5 * max(1.0, abs(float(z["loss_" + split])))


def det_matrix(n, d, c):
    """oracle/gen_golden.py:det_matrix (the deterministic pseudo-embeddings of G17)."""
    i = torch.arange(n, dtype=torch.float64).view(-1, 1)
    j = torch.arange(d, dtype=torch.float64).view(1, -1)
    return (0.6 * torch.sin(0.37 * i + 1.3 * j + c) + 0.4 * torch.cos(0.011 * i * j + 2.0 * c)).float()


def check_post_eval_filters(name, device):
    """PostEvaluationFilter / PostEnsembleEvaluationFilter (utils/post_evaluation.py) against the reference's own ranks (G17):
    exact outside the fp32 tie bands, within the band population inside."""
    from temp_amd.evaluation import PostEnsembleEvaluationFilter, PostEvaluationFilter
    z = load(name)
    s = slice_snapshots()
    D, t, P = int(z["D"]), int(z["t"]), int(z["P"])
    args = make_args(module='GRRGCN', rec_only_last_layer=True, embed_size=D, hidden_size=D, n_bases=8, train_seq_len=4, test_seq_len=4,
                     score_function=str(z["score_function"]))
    m = DynamicRGCN(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(device)
    g = s["va"][t]
    samples = torch.from_numpy(np.stack([g.src, g.rel, g.dst], axis=1)).to(device)
    assert samples.shape[0] == P
    loc, rec = det_matrix(g.n, D, 0.1).to(device), det_matrix(g.n, D, 0.7).to(device)
    all_loc, all_rec = det_matrix(s["num_e"], D, 1.3).to(device), det_matrix(s["num_e"], D, 2.1).to(device)
    rel = (det_matrix(2 * s["num_r"], D, 3.3) * float(z["rel_scale"])).to(device)
    w = [T(z["w%d" % k]).to(device) for k in range(4)]
    post = PostEvaluationFilter(args, m.calc_score, s["tr"], s["va"], s["te"])
    ens = PostEnsembleEvaluationFilter(args, m.calc_score, s["tr"], s["va"], s["te"])
    got = dict(post=post.calc_metrics_single_graph(loc, rec, rel, all_loc, all_rec, samples, w[0], w[1], w[2], w[3], g, t),
               ens=ens.calc_metrics_single_graph(loc, rec, rel, all_loc, all_rec, w[0], w[1], samples, g, t))
    for k, ranks in got.items():
        want, nclose = T(z["ranks_" + k]).long(), T(z["nclose_" + k]).long()
        r = ranks.cpu()
        assert r.shape == want.shape
        safe = nclose == 0
        assert safe.float().mean().item() > 0.85, (name, k)
        assert torch.equal(r[safe], want[safe]), (name, k, int((r[safe] != want[safe]).sum()))
        assert bool(((r - want).abs() <= nclose).all()), (name, k)


def g18_ratio(triples, t, g):
    """oracle/gen_golden.py:g18_ratio (deterministic stand-in for calc_ensemble_ratio)."""
    i = torch.arange(triples.shape[0], dtype=torch.float32, device=triples.device).view(-1, 1)
    return 0.2 + 0.6 * torch.sin(0.7 * i + 0.1) ** 2, 0.25 + 0.5 * torch.cos(0.3 * i) ** 2


def check_post_ensemble_evaluate(name, device, batched=True):
    """evaluate() of the score-level post-ensemble models with an injected calc_ensemble_ratio, against the reference's ranks (G18)."""
    from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN, PostEnsembleDynamicRGCN
    z = load(name)
    bi = str(z["module"]).startswith("Bi")
    m = build_post_model(z, device, PostEnsembleBiDynamicRGCN if bi else PostEnsembleDynamicRGCN, batched, post_ensemble=True)
    with torch.no_grad():
        m.rel_embeds.mul_(float(z["rel_scale"]))
    m.calc_ensemble_ratio = g18_ratio
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    for split, val in (("val", True), ("test", False)):
        ranks, _ = m.evaluate(t_list, val=val)
        want, nclose = T(z["ranks_" + split]).long(), T(z["nclose_" + split]).long()
        got = ranks.cpu()
        assert got.shape == want.shape
        safe = nclose == 0
        assert safe.float().mean().item() > 0.85, (name, split)
        assert torch.equal(got[safe], want[safe]), (name, split, int((got[safe] != want[safe]).sum()))
        assert bool(((got - want).abs() <= nclose).all()), (name, split)


def check_post_ensemble_ratio(name, device, batched=True):
    """PostEnsemble(Bi)DynamicRGCN.forward with the model's OWN calc_ensemble_ratio against the reference's (G19): the frequency
    feature rows fed to the two MLPs (utils/DropEdge.py tables, restated in temp_amd/frequency.py) must be equal, the MLPs carry
    the reference's parameter names and recorded weights, and the training loss with the recorded draws must match.  The
    reference's own backward of this forward() fails under torch 2.x, so gradients are checked for existence only."""
    from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN, PostEnsembleDynamicRGCN
    z = load(name)
    bi = str(z["module"]).startswith("Bi")
    m = build_post_model(z, device, PostEnsembleBiDynamicRGCN if bi else PostEnsembleDynamicRGCN, batched, post_ensemble=True)
    sd = {k[len("mlp_"):]: T(z[k]) for k in z.files if k.startswith("mlp_")}
    assert sorted(sd) == sorted(k for k in m.state_dict() if "_linear." in k) and len(sd) == 8      # the reference's state_dict keys
    m.load_state_dict(sd, strict=False)
    edge_ids, samples = window_inputs(z)
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    wb = m.prepare(t_list, int(z["L"]), True, edge_ids)
    for i, g in enumerate(wb.graphs):
        sub_f, obj_f = m.ensemble_features(samples[i][0], wb.rows[i][-1], g)
        assert torch.equal(sub_f.cpu(), T(z["feat_sub_%d" % i])), (name, i, "subject features")
        assert torch.equal(obj_f.cpu(), T(z["feat_obj_%d" % i])), (name, i, "object features")
    loss = m.run_loss(wb, samples)
    want = float(z["loss"])
    assert abs(loss.item() - want) < 3e-5 * abs(want), (name, loss.item(), want)
    loss.backward()
    assert m.subject_linear[0].weight.grad.abs().sum() > 0 and m.object_linear[2].bias.grad.abs().sum() > 0
    assert m.ent_embeds.grad.abs().sum() > 0
    # evaluate() runs end to end with the model's own weights (no injected function)
    with torch.no_grad():
        ranks, _ = m.evaluate(t_list[:1], val=True)
    assert ranks.numel() > 0 and int(ranks.min()) >= 1


def check_post_ensemble_loss(device):
    """PostEnsembleBiDynamicRGCN.forward with injected mixing weights: the score-level ensemble loss equals its definition
    (models/PostDynamicRGCN.py:399-406) evaluated with the oracle's scorers on the model's own (local, temporal) embeddings."""
    from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN
    z = load("G15_post_bi")
    m = build_post_model(z, device, PostEnsembleBiDynamicRGCN, True, post_ensemble=True)
    edge_ids = [z["choice_%d" % i] for i in range(int(z["n_choices"]))]
    t_list = torch.tensor([int(t) for t in z["t_list"]])
    wb = m.prepare(t_list, int(z["L"]), True, edge_ids)
    gen = torch.Generator().manual_seed(3)
    samples, weights = [], []
    for g in wb.graphs:
        P = min(40, g.number_of_edges())
        trip = torch.from_numpy(np.stack([g.src[:P], g.rel[:P], g.dst[:P]], axis=1))
        samples.append((trip, torch.randint(0, m.num_ents, (P, 6), generator=gen), torch.randint(0, m.num_ents, (P, 6), generator=gen)))
        weights.append((torch.rand(P, 1, generator=gen), torch.rand(P, 1, generator=gen)))
    loss = m.run_loss(wb, samples, weights)
    with torch.no_grad():
        out, hist = m.run(wb)
        recs, locs = list(out.split(wb.target.sizes)), list(wb.out_loc.split(wb.target.sizes))
        want = 0
        for i, g in enumerate(wb.graphs):
            a_loc, a_rec = m.get_all_embeds_Gt(locs[i], recs[i], g, wb.rows[i][-1], wb.plan, i, hist, wb.hist_loc)
            trip, nt, nh = (x.to(device) for x in samples[i])
            ws, wo = (x.to(device) for x in weights[i])
            r = m.rel_embeds[trip[:, 1]]
            lab = torch.zeros(trip.shape[0], dtype=torch.int64, device=device)
            st = wo * O.complex_score(locs[i][trip[:, 0]], r, a_loc[nt], "tail") + (1 - wo) * O.complex_score(recs[i][trip[:, 0]], r, a_rec[nt], "tail")
            # the bidirectional reference class scores the head-corruption candidates as TAILS of the true subject
            # (models/PostBiDynamicRGCN.py:294-295: corrupt_tail=True hard-coded; pinned by G19_post_ratio_bi)
            sh = ws * O.complex_score(locs[i][trip[:, 0]], r, a_loc[nh], "tail") + (1 - ws) * O.complex_score(recs[i][trip[:, 0]], r, a_rec[nh], "tail")
            want = want + torch.nn.functional.cross_entropy(st, lab) + torch.nn.functional.cross_entropy(sh, lab)
    assert abs(loss.item() - want.item()) < 3e-6 * abs(want.item())
    loss.backward()
    assert m.ent_embeds.grad.abs().sum() > 0 and m.ent_encoder.layer_2.forward_rnn.weight_hh_l0.grad.abs().sum() > 0


def check_wide_batched_equals_generic(device, width=260, n_bases=130, module="BiGRRGCN"):
    """Widths the ReLU-folding gather does not take (> 256 columns): layer 2's ReLU adjoint must stay in the layer's own backward
    (the fold flag was decided before the width check, so the mask was applied by neither node).  Batched + chain
    against the reference-granular loop, same seed: outputs and every parameter gradient."""
    s = slice_snapshots()
    t_list = torch.tensor([20, 15, 9, 3])
    res = []
    for batched, chain in ((False, False), (True, True)):
        args = make_args(module=module, rec_only_last_layer=True, embed_size=width, hidden_size=width, n_bases=n_bases, train_seq_len=6,
                         test_seq_len=6)
        torch.manual_seed(13)
        cls = BiDynamicRGCN if module.startswith("Bi") else DynamicRGCN
        m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(device)
        m.use_batched_path, m.use_gru_chain = batched, chain
        m.sample_rng = np.random.default_rng(3)
        per_graph, *_ = m.encode(t_list, 6, True)
        sum((e * e * (i + 1)).sum() for i, e in enumerate(per_graph)).backward()
        res.append(([e.detach().cpu() for e in per_graph], {k: v.grad.detach().cpu().clone() for k, v in m.named_parameters() if v.grad is not None}))
    (o0, g0), (o1, g1) = res
    for a, b in zip(o0, o1):
        assert_close(b, a, 1e-5, 2e-6, "wide batched vs generic")
    assert set(g0) == set(g1) and len(g0) >= 8
    for k in g0:
        assert_close(g1[k], g0[k], 1e-4, 3e-6 * max(1.0, float(g0[k].abs().max())), "wide batched vs generic: d_" + k)


def check_dropout_visits_are_independent(device, module="BiGRRGCN", p=0.1):
    """Self-loop dropout (models/RGCN.py:57-59) draws one mask per encoder call; the reference calls the encoder per window position
    (models/DynamicRGCN.py:156-174), so two windows that visit the same snapshot get DIFFERENT masks.  While dropout draws, no visit
    may share its RGCN rows with another (the batched path used to convolve a shared snapshot once: one mask for all its visits);
    without dropout -- p = 0, or eval mode -- sharing stays on and the results are bit-identical to the shared computation."""
    s = slice_snapshots()
    t_list = torch.tensor([20, 19, 17])                  # overlapping windows: snapshots 12..19 are visited by two or three of them
    L = 8

    def build(pdrop, train_mode):
        args = make_args(module=module, rec_only_last_layer=True, embed_size=32, hidden_size=32, n_bases=16, train_seq_len=L, test_seq_len=L,
                         dropout=pdrop)
        torch.manual_seed(21)
        cls = BiDynamicRGCN if module.startswith("Bi") else DynamicRGCN
        m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(device)
        m.train(train_mode)
        m.sample_rng = np.random.default_rng(3)
        return m

    def visit_rows_of(wb, t):
        """[(first visit row, n)] of every visit of snapshot t among the history steps"""
        out = []
        for st in wb.steps[:-1]:
            off = st.row0
            for g, tt in zip(st.graphs, st.times):
                if tt == t:
                    out.append((off, g.n))
                off += g.n
        return out

    # p = 0: shared; p > 0 in eval mode: shared, bit-identical to p = 0
    m0 = build(0.0, True)
    wb0 = m0.prepare(t_list, L, True)
    assert wb0.shared_visits and wb0.visit_rows_host is not None, "overlapping windows share snapshots"
    out0, _ = m0.run(wb0)
    me = build(p, False)
    wbe = me.prepare(t_list, L, True)
    assert wbe.shared_visits
    oute, _ = me.run(wbe)
    assert torch.equal(out0, oute)
    # p > 0 in training mode: nothing shared, the visits of one snapshot see different masks
    m1 = build(p, True)
    wb1 = m1.prepare(t_list, L, True)
    assert not wb1.shared_visits and wb1.visit_rows_host is None
    assert wb1.n_edges_distinct == wb1.n_edge_visits > wb0.n_edges_distinct
    torch.manual_seed(5)
    out1, _ = m1.run(wb1)
    assert not torch.equal(out1, out0)
    x1, x0 = wb1.last_x.detach(), wb0.last_x.detach()      # GRU input rows (layer-2 outputs) in chain order
    rows1 = wb1.chain_rows_host if getattr(wb1, "chain_rows_host", None) is not None else None
    vis = [v for t in (15, 16, 17) for v in [visit_rows_of(wb1, t)] if len(v) >= 2]
    assert vis, "the batch has snapshots visited by several windows"
    if rows1 is None and x1.shape[0] == wb1.total_rows:      # (uni-directional model: chain order = visit order)
        for v in vis:
            (a, n), (b, _) = v[0], v[1]
            va, vb = x1[a:a + n], x1[b:b + n]
            assert float((va != vb).float().mean()) > 0.3, "two visits of one snapshot must not share a dropout mask"
            assert torch.equal(x0[a:a + n], x0[b:b + n]), "without dropout the visits are the same rows"
    # the dropped run is an unbiased perturbation of the dropout-free one at the layer-1 level; end to end it stays close
    rel = float((out1.detach() - out0.detach()).norm() / out0.detach().norm())
    assert 1e-3 < rel < 0.5, rel
    loss = (out1 * out1).sum()
    loss.backward()
    assert all(torch.isfinite(q.grad).all() for q in m1.parameters() if q.grad is not None)


def check_dropout_visits_self_attention(device, p=0.1):
    """The same rule for the self-attention models (models/SelfAttentionRGCN.py:97-120 encodes every history visit on its own): while
    dropout draws, a history snapshot shared by overlapping windows is encoded once per window."""
    s = slice_snapshots()
    t_list, L = torch.tensor([20, 19, 17]), 6
    rows = {}
    for pdrop, mode in ((0.0, True), (p, False), (p, True)):
        args = make_args(module="SARGCN", rec_only_last_layer=True, embed_size=32, hidden_size=32, n_bases=16, train_seq_len=L, test_seq_len=L,
                         dropout=pdrop, use_time_embedding=True)
        torch.manual_seed(21)
        m = SelfAttentionRGCN(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(device)
        m.train(mode)
        m.sample_rng = np.random.default_rng(3)
        wb = m.prepare(t_list, L, True)
        rows[(pdrop, mode)] = wb.n_hist_rows
        out = m.run(wb)
        assert torch.isfinite(out[0] if isinstance(out, (tuple, list)) else out).all()
    assert rows[(0.0, True)] == rows[(p, False)] < rows[(p, True)]


def check_fused_ensemble_loss(device, head_as_tail):
    """ensemble_loss through the fused node (functional._MixedCandidateCEFn: scores of both streams against ALL entities, mixed on
    the score matrices) against the reference-shaped formulation (gather (P, 1 + neg, D) candidates per stream, mix, CE;
    models/PostDynamicRGCN.py:335-349, 404-406): value and every gradient, incl. the mixing weights'."""
    from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN, PostEnsembleDynamicRGCN
    s = slice_snapshots()
    cls = PostEnsembleBiDynamicRGCN if head_as_tail else PostEnsembleDynamicRGCN
    args = make_args(module="BiGRRGCN" if head_as_tail else "GRRGCN", rec_only_last_layer=True, post_ensemble=True)
    torch.manual_seed(8)
    m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(device)
    assert m.head_scored_as_tail == head_as_tail
    gen = torch.Generator().manual_seed(3)
    N, D, n, P, C = s["num_e"], 32, 60, 37, 21
    mk = lambda *shape: (torch.randn(*shape, generator=gen) * 0.4).to(device).requires_grad_(True)
    res = []
    for fused in (False, True):
        gen.manual_seed(3)
        loc, rec, a_loc, a_rec = mk(n, D), mk(n, D), mk(N, D), mk(N, D)
        ws, wo = torch.sigmoid(mk(P, 1)).detach().requires_grad_(True), torch.sigmoid(mk(P, 1)).detach().requires_grad_(True)
        trip = torch.stack([torch.randint(0, n, (P,), generator=gen), torch.randint(0, 2 * s["num_r"], (P,), generator=gen),
                            torch.randint(0, n, (P,), generator=gen)], dim=1).to(device)
        nt, nh = torch.randint(0, N, (P, C), generator=gen).to(device), torch.randint(0, N, (P, C), generator=gen).to(device)
        m.fused_loss = fused
        m.zero_grad()
        loss = m.ensemble_loss(loc, rec, a_loc, a_rec, trip, nt, nh, ws, wo)
        loss.backward()
        res.append([loss.detach()] + [t.grad.detach().clone() for t in (loc, rec, a_loc, a_rec, ws, wo, m.rel_embeds)])
    for a, b, what in zip(res[0], res[1], ("loss", "d_loc", "d_rec", "d_all_loc", "d_all_rec", "d_w_subject", "d_w_object", "d_rel")):
        assert_close(b, a, 2e-5, 1e-6, "fused ensemble loss: " + what)


def check_static_prepare_split(device):
    """StaticRGCN.prepare + run_loss (the replayable split) == forward() on the same draws: same edge subsamples, same positives
    (same sampler state), the negatives of the forward() call handed to run_loss."""
    s = slice_snapshots()
    args = make_args(module="SRGCN", embed_size=32, hidden_size=32, n_bases=16)
    torch.manual_seed(5)
    m = StaticRGCN(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(device)
    t_list = torch.tensor([20, 15, 9, 3])
    rng = np.random.default_rng(4)
    ids = [np.sort(rng.choice(s["tr"][int(t)].number_of_edges(), s["tr"][int(t)].number_of_edges() // 2, replace=False)) for t in t_list]
    m.sample_rng = np.random.default_rng(9)
    loss = m(t_list, target_edge_ids=ids)
    loss.backward()
    cand = m._last_plan[1]
    want = [loss.detach().clone()] + [p.grad.detach().clone() for p in m.parameters() if p.grad is not None]
    m.zero_grad()
    m.sample_rng = np.random.default_rng(9)
    wb = m.prepare(t_list, ids)
    for _ in range(2):                                   # a prepared batch can be run any number of times
        m.zero_grad()
        l2 = m.run_loss(wb, cand)
        l2.backward()
        got = [l2.detach().clone()] + [p.grad.detach().clone() for p in m.parameters() if p.grad is not None]
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert torch.equal(a, b) or float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))


def check_dropout_all_entity_pass(device, module="BiGRRGCN", rec_only_last_layer=True, p=0.1):
    """The batched all-entity pass (get_all_embeds_Gt of every window at once) while the self-loop dropout draws: the reference runs
    forward_isolated per window, each with its own mask (models/DynamicRGCN.py:56-64, models/RGCN.py:78-89), so the pass keeps one row
    per (window, entity) and ONE launch gives every window its own mask.
      * the (window, entity) layout itself, WITHOUT dropout (`_force_all_rep`): same all-entity matrix, loss and gradients as the
        one-row-per-entity layout;
      * with dropout: the fused pass is taken, rows of an entity that is inactive and never seen in two windows differ between them,
        the loss is finite and stays close to the dropout-free one, every parameter gets a gradient."""
    s = slice_snapshots()
    t_list = torch.tensor([20, 19, 17])
    L = 8

    def build(pdrop):
        args = make_args(module=module, rec_only_last_layer=rec_only_last_layer, embed_size=32, hidden_size=32, n_bases=16, train_seq_len=L,
                         test_seq_len=L, dropout=pdrop, negative_rate=20, num_pos_facts=60)
        torch.manual_seed(21)
        cls = BiDynamicRGCN if module.startswith("Bi") else DynamicRGCN
        m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(device)
        m.train(True)
        m.sample_rng = np.random.default_rng(3)
        m.seed_rng = np.random.default_rng(4)
        return m

    def step(m, force):
        m._force_all_rep = force
        wb = m.prepare(t_list, L, True)
        assert m._fused_all_entity_ok(wb)
        out, hist = m.run(wb)
        big = m.all_embeds_batched(wb, out, hist)
        m.zero_grad()
        loss = m.run_loss(wb)
        loss.backward()
        return wb, big.detach(), float(loss.detach()), {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}

    wb0, big0, loss0, g0 = step(build(0.0), False)
    wb1, big1, loss1, g1 = step(build(0.0), True)
    assert not getattr(wb0, "all_rep", False) and wb1.all_rep
    assert torch.allclose(big0, big1, rtol=1e-5, atol=1e-6), float((big0 - big1).abs().max())
    assert abs(loss0 - loss1) <= 1e-5 * abs(loss0)
    assert set(g0) == set(g1)
    for k in g0:
        err = float((g0[k] - g1[k]).norm() / (g0[k].norm() + 1e-12))
        assert err < 2e-5, (k, err)
    m = build(p)
    torch.manual_seed(9)
    wbp, bigp, lossp, gp = step(m, False)
    assert wbp.all_rep and m._dropout_active()
    B, N = bigp.shape[0], bigp.shape[1]
    act = np.zeros((B, N), dtype=bool)
    for b, g in enumerate(wbp.graphs):
        act[b, g.gids] = True
    plans = wbp.plan if isinstance(wbp.plan, tuple) else (wbp.plan,)
    seen = np.zeros((B, N), dtype=bool)
    for pl in plans:
        seen |= np.stack([pl.final_all(b, L - 1)[0] for b in range(B)]) >= 0
    free = ~act & ~seen                            # entities whose row comes from the table in that window
    both = np.nonzero(free[0] & free[1])[0]
    assert both.size > 3, "the slice has entities outside two windows' graphs and histories"
    same0 = torch.equal(big0[0, both], big0[1, both])
    assert same0, "without dropout those rows are the same in every window"
    frac = float((bigp[0, both] != bigp[1, both]).float().mean())
    assert frac > 0.3, "two windows must not share a dropout mask in the isolated pass (%.2f of the elements differ)" % frac
    assert np.isfinite(lossp) and abs(lossp - loss0) < 0.5 * abs(loss0)
    assert set(gp) == set(g0) and all(torch.isfinite(v).all() for v in gp.values())


def check_post_ensemble_rep_layout(device, bi):
    """The post-ensemble models' batched all-entity pass ((all_loc, all_rec) of every window at once) in the (window, entity) layout it
    takes while the self-loop dropout draws: WITHOUT dropout (`_force_all_rep`) it reproduces the one-row-per-entity layout -- both
    matrices, the ensemble loss and every gradient; WITH dropout the fused pass is still taken and two windows' rows of an entity
    outside both graphs differ in the local stream."""
    from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN, PostEnsembleDynamicRGCN
    from temp_amd.sampling import CorruptTriples
    s = slice_snapshots()
    cls, base = (PostEnsembleBiDynamicRGCN, BiDynamicRGCN) if bi else (PostEnsembleDynamicRGCN, DynamicRGCN)
    t_list, L = torch.tensor([20, 19, 17]), 8

    def build(pdrop):
        args = make_args(module="BiGRRGCN" if bi else "GRRGCN", rec_only_last_layer=True, post_ensemble=True, embed_size=32, hidden_size=32,
                         n_bases=16, train_seq_len=L, test_seq_len=L, dropout=pdrop, negative_rate=20, num_pos_facts=60)
        torch.manual_seed(8)
        m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(device)
        m.train(True)
        m.sample_rng = np.random.default_rng(3)
        m.corrupter = CorruptTriples(m.args, s["tr"], seed=5)
        return m

    def step(m, force):
        m._force_all_rep = force
        wb = m.prepare(t_list, L, True)
        smp = [tuple(x.to(device) for x in t) for t in m.draw_samples(wb)]
        wts = [(torch.full((t[0].shape[0], 1), 0.5, device=device), torch.full((t[0].shape[0], 1), 0.5, device=device)) for t in smp]
        out, hist = m.run(wb)
        both = m.batched_all_embeds_post(wb, out, hist, base)
        assert both is not None
        m.zero_grad()
        loss = m.run_loss(wb, smp, wts)
        loss.backward()
        return wb, both[0].detach(), both[1].detach(), float(loss.detach()), {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}

    wb0, loc0, rec0, l0, g0 = step(build(0.0), False)
    wb1, loc1, rec1, l1, g1 = step(build(0.0), True)
    assert wb1.all_rep and not getattr(wb0, "all_rep", False)
    assert torch.allclose(loc0, loc1, rtol=1e-5, atol=1e-6) and torch.allclose(rec0, rec1, rtol=1e-5, atol=1e-6)
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    assert set(g0) == set(g1)
    for k in g0:
        err = float((g0[k] - g1[k]).norm() / (g0[k].norm() + 1e-12))
        assert err < 2e-5, (k, err)
    wbp, locp, recp, lp, gp = step(build(0.1), False)
    assert wbp.all_rep
    B, N = locp.shape[0], locp.shape[1]
    act = np.zeros((B, N), dtype=bool)
    for b, g in enumerate(wbp.graphs):
        act[b, g.gids] = True
    both_out = np.nonzero(~act[0] & ~act[1])[0]
    assert both_out.size > 3
    assert torch.equal(loc0[0, both_out], loc0[1, both_out])
    assert float((locp[0, both_out] != locp[1, both_out]).float().mean()) > 0.3, "the local stream of two windows must not share a dropout mask"
    assert np.isfinite(lp) and all(torch.isfinite(v).all() for v in gp.values())
