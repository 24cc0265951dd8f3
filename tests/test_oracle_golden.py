"""Pin the CPU oracle (oracle/temp_oracle.py) against golden vectors recorded from the TeMP
reference's own modules (oracle/gen_golden.py).  CPU only; tolerance 1e-5 relative fp32
 with a small absolute floor for near-zero elements."""
import numpy as np
import pytest
import torch

from oracle import temp_oracle as O
from tests.golden_util import T, assert_close, checksum, graph_from, layer_params, load, slice_graphs

RT, AT = 1e-5, 2e-6


def test_G1_msg_func():
    z = load("G1_msg_func")
    for c in range(int(z["ncases"])):
        p = "c%d_" % c
        n = z[p + "h"].shape[0]
        g = O.SnapGraph(n, z[p + "src"], z[p + "dst"], z[p + "rel"], np.arange(n), np.ones(n, np.float32), z[p + "enorm"])
        msg = O.rgcn_messages(T(z[p + "h"]), g, T(z[p + "weight"]), int(z[p + "B"]))
        assert_close(msg, z[p + "msg"], RT, AT, "G1 case %d" % c)


def test_slice_graph_builder_matches_reference():
    z = load("icews14_slice")
    num_e, num_r, times, gd = slice_graphs()
    assert (num_e, num_r) == (7128, 230)
    for t in z["graph_times"]:
        for split in ("train", "valid", "test"):
            p = "g_%s_%d_" % (split, int(t))
            g = gd[split][int(t)]
            assert g.n == int(z[p + "n"])
            for k in ("src", "dst", "rel", "ids"):
                assert np.array_equal(getattr(g, k).numpy(), z[p + k]), (split, t, k)
            assert np.array_equal(g.nnorm.numpy(), z[p + "nnorm"])
            assert np.array_equal(g.enorm.numpy(), z[p + "enorm"])


def test_G2_rgcn_layer_fwd_bwd_and_isolated():
    z = load("G2_rgcn_layer")
    g = graph_from(z)
    for c in range(int(z["ncases"])):
        p = "c%d_" % c
        D, B, bias, act = int(z[p + "D"]), int(z[p + "B"]), bool(z[p + "bias"]), str(z[p + "act"])
        seed = int(z[p + "seed"])
        rng = np.random.default_rng(seed)
        lp = layer_params(rng, D, B, 460, 24, bias)
        ent = O._xavier(rng, 7128, D)
        assert abs(checksum(lp) + ent.double().abs().sum().item() - float(z[p + "param_checksum"])) < 1e-6
        w = lp["weight"].clone().requires_grad_(True)
        lw = lp["loop_weight"].clone().requires_grad_(True)
        b = lp["h_bias"].clone().requires_grad_(True) if bias else None
        h0 = ent[g.ids].clone().requires_grad_(True)
        y = O.rgcn_layer(h0, g, w, lw, B, b, None if act == "none" else act)
        assert_close(y, z[p + "y"], RT, AT, "G2 y case %d" % c)
        gy = T(np.random.default_rng(seed + 1000).standard_normal(tuple(y.shape)).astype(np.float32))
        y.backward(gy)
        assert_close(h0.grad, z[p + "d_h0"], RT, AT, "G2 d_h0 %d" % c)
        assert_close(w.grad[:40], z[p + "d_weight_rows"], RT, AT, "G2 d_weight rows %d" % c)
        assert abs(w.grad.double().abs().sum().item() - float(z[p + "d_weight_abs"])) < 1e-5 * float(z[p + "d_weight_abs"])
        assert_close(lw.grad, z[p + "d_loop"], 2e-5, 1e-5, "G2 d_loop %d" % c)
        if bias:
            assert_close(b.grad, z[p + "d_bias"], 2e-5, 1e-5, "G2 d_bias %d" % c)
        iso = O.rgcn_layer_isolated(ent[:300], lp["loop_weight"], lp["h_bias"], None if act == "none" else act)
        assert_close(iso, z[p + "iso"], RT, AT, "G3 iso %d" % c)
        temb = O.time_embedding_rows(lp["time_embed"], [0, 1, 2, 3], [int(s) for s in z["node_sizes"]])
        assert abs(temb.double().sum().item() - float(z[p + "temb_sum"])) < 1e-6


def _g4_params(z, p, D, B, type1, nl, seed):
    rng = np.random.default_rng(seed)
    lp = layer_params(rng, D, B, 460, 24, False)
    if type1:
        rp = [dict(w_ih=T(z[p + "rnn_w_ih"]), w_hh=T(z[p + "rnn_w_hh"]), b_ih=T(z[p + "rnn_b_ih"]), b_hh=T(z[p + "rnn_b_hh"]))]
        # keep the generator's draw order (4 standard_normal draws) so `ent` matches
        rng.standard_normal((D, D)); rng.standard_normal((3 * D, D)); rng.standard_normal(D); rng.standard_normal(3 * D)
    else:
        rp = O._gru_params(rng, D, nl)
    ent = O._xavier(rng, 7128, D)
    return lp, rp, ent, rng


def test_G4_G5_grrgcn_layer():
    z = load("G4_grrgcn_layer")
    g = graph_from(z)
    for c in range(int(z["ncases"])):
        p = "c%d_" % c
        D, B, type1, learn, nl = int(z[p + "D"]), int(z[p + "B"]), bool(z[p + "type1"]), bool(z[p + "learn"]), int(z[p + "nl"])
        lp, rp, ent, _ = _g4_params(z, p, D, B, type1, nl, int(z[p + "seed"]))
        assert int(z[p + "aliased"]) == 1          # F7: the layer returns the caller's graph object
        leaves = []
        for q in rp:
            for k in q:
                q[k] = q[k].clone().requires_grad_(True)
                leaves.append(q[k])
        layer = dict(weight=lp["weight"], loop_weight=lp["loop_weight"].clone().requires_grad_(True), rnn=rp)
        cfg = dict(n_bases=B, inv_temperature=0.1, type1=type1, learnable_lambda=learn)
        if learn:
            dw = torch.full((1, 1), 0.3, requires_grad=True)
            db = torch.full((1,), -0.2, requires_grad=True)
            layer["exponential_decay"] = (dw, db)
        h0 = ent[g.ids].clone().requires_grad_(True)
        prev = T(z[p + "prev"]).clone().requires_grad_(True)
        dt = T(z[p + "dt"]).view(-1, 1)
        _, hid = O.grrgcn_layer(layer, cfg, g, h0, prev, dt)
        assert_close(hid, z[p + "hid"], RT, AT, "G4 hid %d" % c)
        gy = T(np.random.default_rng(int(z[p + "seed"]) + 1000).standard_normal(tuple(hid.shape)).astype(np.float32))
        hid.backward(gy)
        assert_close(h0.grad, z[p + "d_h0"], 2e-5, 2e-6, "G4 d_h0 %d" % c)
        assert_close(prev.grad, z[p + "d_prev"], 2e-5, 2e-6, "G4 d_prev %d" % c)
        assert_close(layer["loop_weight"].grad, z[p + "d_loop"], 5e-5, 2e-5, "G4 d_loop %d" % c)
        assert_close(rp[0]["w_ih"].grad, z[p + "d_w_ih"], 5e-5, 2e-5, "G4 d_w_ih %d" % c)
        assert_close(rp[0]["w_hh"].grad, z[p + "d_w_hh"], 5e-5, 2e-5, "G4 d_w_hh %d" % c)
        assert_close(rp[0]["b_ih"].grad, z[p + "d_b_ih"], 5e-5, 2e-5, "G4 d_b_ih %d" % c)
        assert_close(rp[0]["b_hh"].grad, z[p + "d_b_hh"], 5e-5, 2e-5, "G4 d_b_hh %d" % c)
        if learn:
            assert_close(dw.grad, z[p + "d_decay_w"], 5e-5, 1e-5, "G4 d_decay_w")
            assert_close(db.grad, z[p + "d_decay_b"], 5e-5, 1e-5, "G4 d_decay_b")


def _enc_model(z, p):
    cfg = dict(module=str(z[p + "module"]), n_bases=int(z[p + "B"]), inv_temperature=0.1,
               rec_only_last_layer=bool(z[p + "rec_only"]), use_time_embedding=bool(z[p + "te"]))
    model = O.init_model(cfg, 7128, 230, 24, int(z[p + "D"]), seed=int(z[p + "seed"]))
    assert abs(checksum(model) - float(z[p + "param_checksum"])) < 1e-6
    return cfg, model


def test_G6_rrgcn_container():
    z = load("G6_rrgcn")
    g = graph_from(z)
    sizes = [int(s) for s in z["node_sizes"]]
    tl = [int(t) for t in z["times"]]
    for c in range(int(z["ncases"])):
        p = "c%d_" % c
        cfg, model = _enc_model(z, p)
        D = int(z[p + "D"])
        rng = np.random.default_rng(int(z[p + "seed"]) + 5000)
        n = g.n
        p1 = T(rng.standard_normal((n, D)).astype(np.float32) * 0.3).requires_grad_(True)
        p2 = T(rng.standard_normal((n, D)).astype(np.float32) * 0.3).requires_grad_(True)
        dt = T(rng.integers(0, 6, (n, 1)).astype(np.float32))
        ent = model["ent_embeds"].clone().requires_grad_(True)
        first, second = O.rrgcn_forward(model["ent_encoder"], cfg, g, ent[g.ids], p1, p2, dt, tl, sizes)
        assert_close(second, z[p + "second"], RT, AT, "G6 second %d" % c)
        assert int(z[p + "same"]) == int(first is second)        # F7 aliasing (GRU module only)
        if first is not second:
            assert_close(first, z[p + "first"], RT, AT, "G6 first %d" % c)
        gy = T(np.random.default_rng(int(z[p + "seed"]) + 1000).standard_normal(tuple(second.shape)).astype(np.float32))
        (second * gy).sum().backward()
        d1 = p1.grad if p1.grad is not None else torch.zeros_like(p1)
        assert_close(d1, z[p + "d_p1"], 3e-5, 3e-6, "G6 d_p1 %d" % c)
        assert_close(p2.grad, z[p + "d_p2"], 3e-5, 3e-6, "G6 d_p2 %d" % c)
        assert_close(ent.grad[g.ids], z[p + "d_ent_rows"], 3e-5, 3e-6, "G6 d_ent %d" % c)
        iso = O.rrgcn_isolated(model["ent_encoder"], cfg, model["ent_embeds"][:256], T(z[p + "iso_q1"]), T(z[p + "iso_q2"]),
                               T(z[p + "iso_dt"]).view(-1, 1), 8)
        assert_close(iso, z[p + "iso"], RT, AT, "G6 iso %d" % c)
        if cfg["module"] == "GRRGCN":
            loc, f2, s2 = O.rrgcn_forward(model["ent_encoder"], cfg, g, model["ent_embeds"][g.ids], p1.detach(), p2.detach(),
                                          dt, tl, sizes, post=True)
            assert_close(loc, z[p + "post_loc"], RT, AT, "G8 post local %d" % c)
            assert_close(s2, z[p + "post_second"], RT, AT, "G8 post second %d" % c)


def test_G7_birrgcn_container():
    z = load("G7_birrgcn")
    g = graph_from(z)
    sizes = [int(s) for s in z["node_sizes"]]
    tl = [int(t) for t in z["times"]]
    for c in range(int(z["ncases"])):
        p = "c%d_" % c
        cfg, model = _enc_model(z, p)
        D = int(z[p + "D"])
        rng = np.random.default_rng(int(z[p + "seed"]) + 5000)
        n = g.n
        mk = lambda: T(rng.standard_normal((n, D)).astype(np.float32) * 0.3).requires_grad_(True)
        f1, f2, b1, b2 = mk(), mk(), mk(), mk()
        dtf = T(rng.integers(0, 6, (n, 1)).astype(np.float32))
        dtb = T(rng.integers(0, 6, (n, 1)).astype(np.float32))
        ent = model["ent_embeds"].clone().requires_grad_(True)
        second = O.birrgcn_forward(model["ent_encoder"], cfg, g, ent[g.ids], f1, f2, dtf, b1, b2, dtb, tl, sizes)
        assert_close(second, z[p + "second"], RT, AT, "G7 second %d" % c)
        gy = T(np.random.default_rng(int(z[p + "seed"]) + 1000).standard_normal(tuple(second.shape)).astype(np.float32))
        (second * gy).sum().backward()
        zg = lambda q: q.grad if q.grad is not None else torch.zeros_like(q)
        for nm, q in (("d_f1", f1), ("d_f2", f2), ("d_b1", b1), ("d_b2", b2)):
            assert_close(zg(q), z[p + nm], 3e-5, 3e-6, "G7 %s %d" % (nm, c))
        assert_close(ent.grad[g.ids], z[p + "d_ent_rows"], 3e-5, 3e-6, "G7 d_ent %d" % c)
        e0 = model["ent_embeds"][g.ids]
        for fwd in (True, False):
            a, b = O.birrgcn_forward_one_direction(model["ent_encoder"], cfg, g, e0, f1.detach(), f2.detach(), dtf, fwd, tl, sizes)
            tag = "fwd" if fwd else "bwd"
            assert_close(b, z[p + "one_" + tag], RT, AT, "G7 one_direction %s %d" % (tag, c))
            assert int(z[p + "one_same_" + tag]) == int(a is b)
            if a is not b:
                assert_close(a, z[p + "one_first_" + tag], RT, AT, "G7 one_direction first %s %d" % (tag, c))
        iso = O.birrgcn_isolated(model["ent_encoder"], cfg, model["ent_embeds"][:256], T(z[p + "iso_f1"]), T(z[p + "iso_f2"]),
                                 T(z[p + "iso_dtf"]).view(-1, 1), T(z[p + "iso_b1"]), T(z[p + "iso_b2"]),
                                 T(z[p + "iso_dtb"]).view(-1, 1), 10)
        assert_close(iso, z[p + "iso"], RT, AT, "G7 iso %d" % c)
        if cfg["module"] == "BiGRRGCN":
            loc, s2 = O.birrgcn_forward(model["ent_encoder"], cfg, g, e0, f1.detach(), f2.detach(), dtf, b1.detach(), b2.detach(),
                                        dtb, tl, sizes, post=True)
            assert_close(loc, z[p + "post_loc"], RT, AT, "G8 bi post local %d" % c)
            assert_close(s2, z[p + "post_second"], RT, AT, "G8 bi post second %d" % c)


def test_G9_scorers():
    z = load("G9_scores")
    s, r, o, cand = T(z["s"]), T(z["r"]), T(z["o"]), T(z["cand"])
    for name, fn in O.SCORERS.items():
        assert_close(fn(s, r, o), z[name + "_single"], RT, AT, name)
        assert_close(fn(s, r, cand, mode="tail"), z[name + "_tail"], RT, AT, name)
        assert_close(fn(cand, r, o, mode="head"), z[name + "_head"], RT, AT, name)


def window_inputs(z, gd_train):
    """Rebuild the injected random draws of a G10 fixture: subsampled target graphs + negatives."""
    tl = sorted([int(t) for t in z["t_list"]], reverse=True)
    assert int(z["n_choices"]) == len(tl) and int(z["n_samples"]) == len(tl)
    targets = [O.edge_subgraph(gd_train[t], z["choice_%d" % i]) for i, t in enumerate(tl)]
    samples = [(T(z["trip_%d" % i]).long(), T(z["negtail_%d" % i]).long(), T(z["neghead_%d" % i]).long()) for i in range(len(tl))]
    return tl, targets, samples


@pytest.mark.parametrize("name", ["G10_uni_grrgcn", "G10_uni_grrgcn_rol", "G10_bi_grrgcn_rol", "G10_bi_grrgcn",
                                  "G10_uni_grrgcn_d200", "G10_bi_grrgcn_rol_d200"])
def test_G10_window_loss_and_grads(name):
    z = load(name)
    num_e, num_r, times, gd = slice_graphs()
    cfg = dict(module=str(z["module"]), n_bases=int(z["B"]), inv_temperature=0.1, rec_only_last_layer=bool(z["rec_only"]),
               use_time_embedding=bool(z["te"]))
    model = O.init_model(cfg, num_e, num_r, len(times), int(z["D"]), seed=int(z["seed"]))
    assert abs(checksum(model) - float(z["param_checksum"])) < 1e-6
    leaves = O.leaf_tensors(model)
    for v in leaves.values():
        v.requires_grad_(True)
    tl, targets, samples = window_inputs(z, gd["train"])
    L = int(z["L"])
    fn = O.bi_forward_loss if cfg["module"].startswith("Bi") else O.uni_forward_loss
    loss, _ = fn(model, cfg, gd["train"], tl, times, L, targets, samples)
    assert abs(loss.item() - float(z["loss"])) < 2e-5 * abs(float(z["loss"]))
    loss.backward()
    eg = model["ent_embeds"].grad
    rows = T(z["d_ent_nz_rows"]).long()
    assert_close(eg[rows], z["d_ent_nz_vals"], 1e-4, 2e-6, name + " d_ent")
    if "d_ent_sub" in z.files:              # sub-sampled rows: the global sums pin the rest
        want = float(z["gabs_ent_embeds"])
        assert abs(eg.double().abs().sum().item() - want) < 2e-4 * want
    else:
        mask = torch.ones(eg.shape[0], dtype=torch.bool)
        mask[rows] = False
        assert float(eg[mask].abs().max()) < 1e-7 if mask.any() else True
    assert_close(model["rel_embeds"].grad, z["d_rel"], 1e-4, 2e-6, name + " d_rel")
    key_map = {"w_ih": "weight_ih_l0", "w_hh": "weight_hh_l0", "b_ih": "bias_ih_l0", "b_hh": "bias_hh_l0"}
    checked = 0
    for k, v in leaves.items():
        if v.grad is None or not k.startswith("ent_encoder"):
            continue
        parts = k.split(".")
        if parts[-1] in key_map:                 # ent_encoder.layer_2.rnn.0.w_ih -> ...rnn.weight_ih_l0
            ref_key = ".".join(parts[:-2] + [key_map[parts[-1]]])
        else:
            ref_key = k
        gk = "gabs_" + ref_key
        if gk in z.files:
            want = float(z[gk])
            assert abs(v.grad.double().abs().sum().item() - want) < 2e-4 * max(want, 1e-3), (name, k)
            checked += 1
    assert checked >= 4


def test_G11_history_trace_semantics():
    """F8: history re-zeroed every step; start_time persists; F7: both slots equal."""
    z = load("G10_uni_grrgcn")
    num_e, num_r, times, gd = slice_graphs()
    cfg = dict(module="GRRGCN", n_bases=int(z["B"]), inv_temperature=0.1, rec_only_last_layer=False, use_time_embedding=False)
    model = O.init_model(cfg, num_e, num_r, len(times), int(z["D"]), seed=int(z["seed"]))
    tl = sorted([int(t) for t in z["t_list"]], reverse=True)
    L = int(z["L"])
    tbl = O.get_batch_graph_list(tl, L, times)
    ent = model["ent_embeds"]
    H = O.DenseHistory(len(tl), num_e, ent.shape[1], ent.dtype)
    i = 0
    for cur_t in range(L - 1):
        ts = [t for t in tbl[cur_t] if t is not None]
        if not ts:
            continue
        graphs = [gd["train"][t] for t in ts]
        sizes = [g.n for g in graphs]
        fp, sp, dt = H.get_prev(graphs, cur_t)
        bg = O.batch_graphs(graphs)
        first, second = O.rrgcn_forward(model["ent_encoder"], cfg, bg, ent[bg.ids], fp, sp, dt, tbl[cur_t], sizes)
        H.update(first.split(sizes), second.split(sizes), graphs, cur_t)
        assert int(z["tr%d_cur_t" % i]) == cur_t
        for b in range(len(tl)):
            rows = T(z["tr%d_b%d_rows" % (i, b)]).long()
            got_rows = torch.nonzero(H.hist[b, 1].abs().sum(1)).view(-1)
            assert torch.equal(rows, got_rows)
            assert_close(H.hist[b, 1][rows], z["tr%d_b%d_vals" % (i, b)], 2e-5, 2e-6, "trace step %d b %d" % (i, b))
            assert int(z["tr%d_b%d_same" % (i, b)]) == 1 and torch.equal(H.hist[b, 0], H.hist[b, 1])
            srows = T(z["tr%d_b%d_srows" % (i, b)]).long()
            assert torch.equal(srows, torch.nonzero(H.start[b]).view(-1))
            assert_close(H.start[b][srows], z["tr%d_b%d_svals" % (i, b)], 0, 0, "start")
        i += 1
    assert i == int(z["n_trace"])


def test_G12_static_rgcn():
    z = load("G12_static_rgcn")
    num_e, num_r, times, gd = slice_graphs()
    D, B, seed = int(z["D"]), int(z["B"]), int(z["seed"])
    cfg = dict(module="SRGCN", n_bases=B, inv_temperature=0.1, rec_only_last_layer=False, use_time_embedding=False)
    model = O.init_model(cfg, num_e, num_r, len(times), D, seed=seed, bias=True)
    rng = np.random.default_rng(seed + 1)
    for ln in ("layer_1", "layer_2"):
        model["ent_encoder"][ln]["h_bias"] = T(rng.uniform(-0.3, 0.3, D).astype(np.float32))
    assert abs(checksum(model) - float(z["param_checksum"])) < 1e-6
    tl = [int(t) for t in z["t_list"]]
    embeds = O.static_forward_embeds(model, cfg, [gd["train"][t] for t in tl], tl)
    for i, e in enumerate(embeds):
        assert_close(e, z["emb_%d" % i], RT, AT, "G12 emb %d" % i)
    iso = O.static_rgcn_isolated(model["ent_encoder"], cfg, model["ent_embeds"][:200], tl[0])
    assert_close(iso, z["iso"], RT, AT, "G12 iso")
    # the training step: loss + gradients with the reference's recorded draws
    leaves = O.leaf_tensors(model)
    for v in leaves.values():
        v.requires_grad_(True)
    targets = [O.edge_subgraph(gd["train"][t], z["choice_%d" % i]) for i, t in enumerate(tl)]
    samples = [(T(z["trip_%d" % i]).long(), T(z["negtail_%d" % i]).long(), T(z["neghead_%d" % i]).long()) for i in range(len(tl))]
    loss, _ = O.static_forward_loss(model, cfg, gd["train"], tl, targets, samples)
    assert abs(loss.item() - float(z["loss"])) < 2e-5 * abs(float(z["loss"]))
    loss.backward()
    eg = model["ent_embeds"].grad
    rows = T(z["d_ent_nz_rows"]).long()
    assert_close(eg[rows], z["d_ent_nz_vals"], 1e-4, 2e-6, "G12 d_ent")
    assert_close(model["rel_embeds"].grad, z["d_rel"], 1e-4, 2e-6, "G12 d_rel")
    for ln in ("layer_1", "layer_2"):
        assert_close(model["ent_encoder"][ln]["h_bias"].grad, z["d_bias_" + ln], 1e-4, 2e-6, "G12 d_bias")


@pytest.mark.parametrize("name", ["G13_eval_uni", "G13_eval_bi"])
def test_G13_filtered_ranks(name):
    """The oracle's restatement of evaluate() (window encoder on the full train graphs -> all-entity matrix -> filtered
    ranks, utils/evaluation.py:34-106) against the reference's own ranks: exact wherever no competitor sits inside the fp32
    tie band of the target (see oracle/gen_golden.py:gen_G13), within the band population elsewhere."""
    z = load(name)
    num_e, num_r, times, gd = slice_graphs()
    cfg = dict(module=str(z["module"]), n_bases=int(z["B"]), inv_temperature=0.1, rec_only_last_layer=bool(z["rec_only"]),
               use_time_embedding=False)
    model = O.init_model(cfg, num_e, num_r, len(times), int(z["D"]), seed=int(z["seed"]))
    model["rel_embeds"] = model["rel_embeds"] * float(z["rel_scale"])
    tl = sorted([int(t) for t in z["t_list"]], reverse=True)
    L = int(z["L"])
    bi = cfg["module"].startswith("Bi")
    with torch.no_grad():
        targets = [gd["train"][t] for t in tl]
        if bi:
            tf, tb = O.get_batch_graph_list_bi(tl, L, times)
            Hf = O.bi_pre_forward(model, cfg, gd["train"], tf, L, True)
            Hb = O.bi_pre_forward(model, cfg, gd["train"], tb, L, False)
            per_graph = O.bi_target_embeds(model, cfg, Hf, Hb, targets, tf[-1], L)
        else:
            tf = O.get_batch_graph_list(tl, L, times)
            H = O.uni_pre_forward(model, cfg, gd["train"], tf, L)
            per_graph = O.uni_target_embeds(model, cfg, H, targets, tf[-1], L)
        for split in ("val", "test"):
            gsplit = gd["valid" if split == "val" else "test"]
            ranks = []
            for i, t in enumerate(tl):
                g = gsplit[t]
                if g.num_edges == 0:
                    continue
                emb = per_graph[i]
                all_e = (O.bi_all_embeds(model, cfg, Hf, Hb, i, g, t, emb, L) if bi else O.uni_all_embeds(model, cfg, H, i, g, t, emb, L))
                trip = torch.from_numpy(np.stack([g.src, g.rel, g.dst], axis=1))
                heads, tails = O.true_heads_and_tails([np.stack([x[t].src, x[t].rel, x[t].dst], axis=1) for x in (gd["train"], gd["valid"], gd["test"])])
                ranks.append(O.filtered_ranks(O.complex_score, emb, model["rel_embeds"], all_e, trip, g.ids, tails, heads))
            got = torch.cat(ranks)
            want, nclose = T(z["ranks_" + split]).long(), T(z["nclose_" + split]).long()
            assert got.shape == want.shape
            safe = nclose == 0
            assert torch.equal(got[safe], want[safe]), (name, split)
            assert bool(((got - want).abs() <= nclose).all()), (name, split)


@pytest.mark.parametrize("name", ["G14_sa_uni_rol", "G14_sa_uni", "G14_sa_bi_rol"])
def test_G14_self_attention_window_loss_and_grads(name):
    """Config 5 (SARGCN attention over the history window) against the reference's own
    SelfAttentionRGCN / BiSelfAttentionRGCN forward + backward."""
    z = load(name)
    num_e, num_r, times, gd = slice_graphs()
    cfg = dict(module=str(z["module"]), n_bases=int(z["B"]), inv_temperature=0.1, rec_only_last_layer=bool(z["rec_only"]),
               use_time_embedding=True, learnable_lambda=bool(z["learn"]))
    model = O.init_model(cfg, num_e, num_r, len(times), int(z["D"]), seed=int(z["seed"]))
    if cfg["learnable_lambda"]:
        for ln in ("layer_1", "layer_2"):
            model["ent_encoder"][ln]["exponential_decay"] = (torch.full((1, 1), 0.25), torch.full((1,), -0.1))
    assert abs(checksum(model) - float(z["param_checksum"])) < 1e-6
    leaves = O.leaf_tensors(model)
    for v in leaves.values():
        v.requires_grad_(True)
    tl, targets, samples = window_inputs(z, gd["train"])
    loss, _ = O.sa_forward_loss(model, cfg, gd["train"], tl, times, int(z["L"]), targets, samples, bi=cfg["module"].startswith("Bi"))
    assert abs(loss.item() - float(z["loss"])) < 2e-5 * abs(float(z["loss"]))
    loss.backward()
    eg = model["ent_embeds"].grad
    rows = T(z["d_ent_nz_rows"]).long()
    assert_close(eg[rows], z["d_ent_nz_vals"], 1e-4, 2e-6, name + " d_ent")
    assert_close(model["rel_embeds"].grad, z["d_rel"], 1e-4, 2e-6, name + " d_rel")
    checked = 0
    for k, v in leaves.items():
        if v.grad is None or not k.startswith("ent_encoder"):
            continue
        parts = k.split(".")
        ref_key = k + ".weight" if parts[-1] in ("q_linear", "k_linear", "v_linear") else k
        if parts[-2] == "exponential_decay":
            ref_key = ".".join(parts[:-1] + ["weight" if parts[-1] == "0" else "bias"])
        gk = "gabs_" + ref_key
        if gk in z.files:
            want = float(z[gk])
            assert abs(v.grad.double().abs().sum().item() - want) < 2e-4 * max(want, 1e-3), (name, k, want)
            checked += 1
    assert checked >= 8, checked


# --------------------------------------------------------------------------------------
# a22 / config 3: post-ensemble and impute window models (G15)
# --------------------------------------------------------------------------------------
IMPUTE_GATES = {"impute_weight": (0.3, -0.1), "impute_weight_forward": (0.25, -0.05), "impute_weight_backward": (0.4, 0.1)}


def _with_impute_gates(model, bi):
    names = ("impute_weight_forward", "impute_weight_backward") if bi else ("impute_weight",)
    for nm in names:
        w, b = IMPUTE_GATES[nm]
        model["ent_encoder"][nm] = (torch.full((1, 1), w), torch.full((1,), b))
    return model


def test_G15_post_ensemble_bi_window():
    """Config 3's flags (BiGRRGCN, --rec-only-last-layer, --post-ensemble, L = 15): (local, temporal) target embeddings,
    local history streams, all-entity (local, temporal) matrices and the gradients of their seeded weighted sum."""
    z = load("G15_post_bi")
    num_e, num_r, times, gd = slice_graphs()
    cfg = dict(module="BiGRRGCN", n_bases=int(z["B"]), inv_temperature=0.1, rec_only_last_layer=True, use_time_embedding=False)
    model = O.init_model(cfg, num_e, num_r, len(times), int(z["D"]), seed=int(z["seed"]))
    assert abs(checksum(model) - float(z["param_checksum"])) < 1e-6
    leaves = O.leaf_tensors(model)
    for v in leaves.values():
        v.requires_grad_(True)
    tl = sorted([int(t) for t in z["t_list"]], reverse=True)
    L = int(z["L"])
    targets = [O.edge_subgraph(gd["train"][t], z["choice_%d" % i]) for i, t in enumerate(tl)]
    tf, tb = O.get_batch_graph_list_bi(tl, L, times)
    Hf = O.post_bi_pre_forward(model, cfg, gd["train"], tf, L, True)
    Hb = O.post_bi_pre_forward(model, cfg, gd["train"], tb, L, False)
    loc, rec = O.post_bi_target_embeds(model, cfg, Hf, Hb, targets, tf[-1], L)
    rows_sel = T(z["all_rows"]).long()
    gen = torch.Generator().manual_seed(int(z["seed"]))
    total = 0
    for i, t in enumerate(tl):
        assert_close(loc[i], z["loc_%d" % i], RT, AT, "G15 loc %d" % i)
        assert_close(rec[i], z["rec_%d" % i], RT, AT, "G15 rec %d" % i)
        for nm, H in (("f_loc", Hf), ("b_loc", Hb)):
            rows = T(z["%s_%d_rows" % (nm, i)]).long()
            assert torch.equal(rows, torch.nonzero(H.loc[i].detach().abs().sum(1)).view(-1))
            assert_close(H.loc[i][rows], z["%s_%d_vals" % (nm, i)], RT, AT, "G15 %s %d" % (nm, i))
        a_loc, a_rec = O.post_bi_all_embeds(model, cfg, Hf, Hb, i, t, L)
        assert_close(a_loc[rows_sel], z["all_loc_%d" % i], RT, AT, "G15 all loc")
        assert_close(a_rec[rows_sel], z["all_rec_%d" % i], RT, AT, "G15 all rec")
        for x in (loc[i], rec[i], a_loc[rows_sel], a_rec[rows_sel]):
            total = total + (x * torch.randn(x.shape, generator=gen)).sum()
    assert abs(total.item() - float(z["total"])) < 2e-5 * max(1.0, abs(float(z["total"])))
    total.backward()
    eg = model["ent_embeds"].grad
    assert_close(eg[T(z["d_ent_nz_rows"]).long()], z["d_ent_nz_vals"], 1e-4, 2e-6, "G15 d_ent")
    assert abs(eg.double().abs().sum().item() - float(z["gabs_ent_embeds"])) < 2e-4 * float(z["gabs_ent_embeds"])


@pytest.mark.parametrize("name", ["G15_impute_bi", "G15_impute_uni", "G15_impute_uni_full"])
def test_G15_impute_window_loss_and_grads(name):
    z = load(name)
    num_e, num_r, times, gd = slice_graphs()
    bi = str(z["module"]).startswith("Bi")
    cfg = dict(module=str(z["module"]), n_bases=int(z["B"]), inv_temperature=0.1, rec_only_last_layer=bool(z["rec_only"]),
               use_time_embedding=False, impute=True)
    model = O.init_model(cfg, num_e, num_r, len(times), int(z["D"]), seed=int(z["seed"]))
    assert abs(checksum(model) - float(z["param_checksum"])) < 1e-6
    _with_impute_gates(model, bi)
    leaves = O.leaf_tensors(model)
    for v in leaves.values():
        v.requires_grad_(True)
    tl, targets, samples = window_inputs(z, gd["train"])
    fn = O.impute_bi_forward_loss if bi else O.impute_uni_forward_loss
    loss, _ = fn(model, cfg, gd["train"], tl, times, int(z["L"]), targets, samples)
    assert abs(loss.item() - float(z["loss"])) < 2e-5 * abs(float(z["loss"]))
    loss.backward()
    eg = model["ent_embeds"].grad
    assert_close(eg[T(z["d_ent_nz_rows"]).long()], z["d_ent_nz_vals"], 1e-4, 2e-6, name + " d_ent")
    assert_close(model["rel_embeds"].grad, z["d_rel"], 1e-4, 2e-6, name + " d_rel")
    for nm in (("impute_weight_forward", "impute_weight_backward") if bi else ("impute_weight",)):
        w, b = model["ent_encoder"][nm]
        assert abs(w.grad.abs().sum().item() - float(z["gabs_ent_encoder.%s.weight" % nm])) < 2e-4 * max(float(z["gabs_ent_encoder.%s.weight" % nm]), 1e-3)
        assert abs(b.grad.abs().sum().item() - float(z["gabs_ent_encoder.%s.bias" % nm])) < 2e-4 * max(float(z["gabs_ent_encoder.%s.bias" % nm]), 1e-3)
