#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the TeMP reference's OWN modules.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (needs /root/reference, read-only);
imports the reference in place under `oracle/ref_stubs` (dgl / pytorch_lightning stand-ins,
see their docstrings) and records inputs + outputs as data.  No reference source is copied.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py            # all fixtures
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py G2 G10     # a subset

Parameters are NOT drawn from the reference's initialisers: they come from
`oracle.temp_oracle.init_model(seed)` (numpy PCG64, same shapes/ranges) and are loaded into the
reference modules, so fixtures only need to carry the seed plus a checksum.  Random choices the
reference makes from unseeded np.random (edge subsample, negatives -- SURVEY F11) are captured
and stored as inputs.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import ref_harness as rh  # noqa: E402
from oracle import temp_oracle as O  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
warnings.filterwarnings("ignore")
N_TIMES = 24


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    clean = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        clean[k] = np.asarray(v)
    np.savez_compressed(path, **clean)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def checksum(model):
    return float(sum(v.double().abs().sum().item() for v in O.leaf_tensors(model).values()))


def graph_arrays(g, prefix=""):
    """reference DGL(stub) graph -> plain arrays (the hot path's input contract)."""
    src, dst = g.edges()
    d = {prefix + "n": g.number_of_nodes(), prefix + "src": src.numpy(), prefix + "dst": dst.numpy(),
         prefix + "rel": g.edata['type_s'].numpy(), prefix + "ids": g.ndata['id'].view(-1).numpy(),
         prefix + "nnorm": g.ndata['norm'].view(-1).numpy().astype(np.float32)}
    if 'norm' in g.edata:
        d[prefix + "enorm"] = g.edata['norm'].view(-1).numpy().astype(np.float32)
    return d


def to_ref_state_dict(model, type1=False):
    """oracle parameter dict -> reference state_dict keys."""
    sd = {'ent_embeds': model['ent_embeds'], 'rel_embeds': model['rel_embeds']}
    for ln, d in model['ent_encoder'].items():
        p = 'ent_encoder.%s.' % ln
        for k in ('weight', 'loop_weight', 'time_embed', 'time_weight', 'time_weight_forward', 'time_weight_backward'):
            if d.get(k) is not None:
                sd[p + k] = d[k]
        if d.get('h_bias') is not None:
            sd[p + 'h_bias'] = d['h_bias']
        for name in ('q_linear', 'k_linear', 'v_linear'):
            if name in d:
                sd[p + name + '.weight'] = d[name]
        for name in ('rnn', 'forward_rnn', 'backward_rnn'):
            if name in d:
                for li, q in enumerate(d[name]):
                    suf = '' if type1 else '_l%d' % li
                    sd[p + name + '.weight_ih' + suf] = q['w_ih']
                    sd[p + name + '.weight_hh' + suf] = q['w_hh']
                    sd[p + name + '.bias_ih' + suf] = q['b_ih']
                    sd[p + name + '.bias_hh' + suf] = q['b_hh']
        if d.get('exponential_decay') is not None:
            sd[p + 'exponential_decay.weight'] = d['exponential_decay'][0]
            sd[p + 'exponential_decay.bias'] = d['exponential_decay'][1]
    return sd


_CACHE = {}


def graphs():
    if 'g' not in _CACHE:
        _CACHE['g'] = rh.build_graph_dicts(max_times=N_TIMES)
    return _CACHE['g']


# ----------------------------------------------------------------------------------------
def gen_slice():
    """The 'identical ICEWS14 inputs': quads of the first N_TIMES timestamps (public dataset
    text, data not code) + the per-timestamp graphs the reference builds from them."""
    from utils.dataset import load_quadruples
    out = {}
    tmax = None
    for split in ('train', 'valid', 'test'):
        q, times = load_quadruples('interpolation/icews14', split + '.txt')
        if tmax is None:
            _, all_t = load_quadruples('interpolation/icews14', 'train.txt', 'valid.txt', 'test.txt')
            tmax = all_t[N_TIMES - 1]
            out['times'] = all_t[:N_TIMES]
        out[split] = q[q[:, 3] <= tmax].astype(np.int32)
    num_e, num_r, tr, va, te = graphs()
    out['num_ents'], out['num_rels'] = num_e, num_r
    tkeys = list(tr.keys())
    out['graph_times'] = np.array([int(tkeys[i]) for i in (0, 3, 20)])
    for i in (0, 3, 20):
        t = tkeys[i]
        for nm, gd in (('train', tr), ('valid', va), ('test', te)):
            out.update(graph_arrays(gd[t], 'g_%s_%d_' % (nm, int(t))))
    save("icews14_slice", **out)


def gen_G1():
    """RGCNLayer.msg_func (models/RGCN.py:91-98)."""
    from models.RGCN import RGCNLayer
    import dgl
    args = rh.make_args()
    rng = np.random.default_rng(11)
    out = {}
    for ci, (D, B) in enumerate([(200, 100), (128, 128), (8, 2), (16, 4)]):
        R2, n, E = 12, 40, 64
        layer = RGCNLayer(args, D, D, R2, B, list(range(4)), bias=False, activation=None, self_loop=True, dropout=0.0)
        w = O._xavier(rng, R2, layer.weight.shape[1])
        layer.weight.data.copy_(w)
        g = dgl.DGLGraph()
        g.add_nodes(n)
        src, dst = rng.integers(0, n, E), rng.integers(0, n, E)
        g.add_edges(src, dst)
        h = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32))
        rel = torch.from_numpy(rng.integers(0, R2, E))
        en = torch.from_numpy(rng.uniform(0.1, 1.0, (E, 1)).astype(np.float32))
        g.ndata['h'] = h
        g.edata['type_s'] = rel
        g.edata['norm'] = en
        msg = layer.msg_func(g._edge_batch())['msg']
        pre = 'c%d_' % ci
        out.update({pre + 'D': D, pre + 'B': B, pre + 'h': h, pre + 'src': src, pre + 'dst': dst, pre + 'rel': rel,
                    pre + 'enorm': en.view(-1), pre + 'weight': w, pre + 'msg': msg})
    save("G1_msg_func", ncases=4, **out)


def _layer_params(rng, D, B, R2, T, bias):
    s = D // B
    return dict(weight=O._xavier(rng, R2, B * s * s), loop_weight=O._xavier(rng, D, D), time_embed=O._xavier(rng, T, D),
                h_bias=(torch.from_numpy(rng.uniform(-0.5, 0.5, D).astype(np.float32)) if bias else None))


def _set_layer(layer, p):
    layer.weight.data.copy_(p['weight'])
    layer.loop_weight.data.copy_(p['loop_weight'])
    layer.time_embed.data.copy_(p['time_embed'])
    if p.get('h_bias') is not None:
        layer.h_bias.data.copy_(p['h_bias'])


def gen_G2_G3():
    """RGCNLayer.forward / forward_isolated on real ICEWS14 snapshots (models/RGCN.py:53-89)."""
    import dgl
    import torch.nn.functional as F
    from models.RGCN import RGCNLayer
    num_e, num_r, tr, va, te = graphs()
    times = list(tr.keys())
    args = rh.make_args()
    gl = [tr[times[i]] for i in range(4)]
    bg = dgl.batch(gl)
    sizes = [g.number_of_nodes() for g in gl]
    out = graph_arrays(bg)
    out['node_sizes'] = sizes
    out['times'] = [0, 1, 2, 3]
    case = 0
    for (D, B) in [(200, 100), (32, 32), (16, 4)]:
        for bias in (False, True):
            for act in (None, 'relu'):
                if D == 200 and (bias != (act == 'relu')):
                    continue            # keep the D=200 fixtures small: (no bias, no act) and (bias, relu)
                seed = 100 + case
                rng = np.random.default_rng(seed)
                p = _layer_params(rng, D, B, 2 * num_r, len(times), bias)
                ent = O._xavier(rng, num_e, D)
                layer = RGCNLayer(args, D, D, 2 * num_r, B, times, bias=bias, activation=(F.relu if act else None),
                                  self_loop=True, dropout=0.0)
                _set_layer(layer, p)
                h0 = ent[bg.ndata['id'].view(-1)].clone().requires_grad_(True)
                bg.ndata['h'] = h0
                rg, temb = layer(bg, [0, 1, 2, 3], sizes)
                y = rg.ndata['h']
                gy = torch.from_numpy(np.random.default_rng(seed + 1000).standard_normal(tuple(y.shape)).astype(np.float32))
                y.backward(gy)
                iso, t_iso = layer.forward_isolated(ent[:300].clone(), 2)
                pre = 'c%d_' % case
                out.update({pre + 'D': D, pre + 'B': B, pre + 'bias': int(bias), pre + 'act': (act or 'none'),
                            pre + 'seed': seed, pre + 'y': y, pre + 'temb_sum': temb.double().sum().item(),
                            pre + 'd_h0': h0.grad, pre + 'd_weight_rows': layer.weight.grad[:40],
                            pre + 'd_weight_sum': layer.weight.grad.double().sum().item(),
                            pre + 'd_weight_abs': layer.weight.grad.double().abs().sum().item(),
                            pre + 'd_loop': layer.loop_weight.grad,
                            pre + 'iso': iso, pre + 'param_checksum': checksum(p) + ent.double().abs().sum().item()})
                if bias:
                    out[pre + 'd_bias'] = layer.h_bias.grad
                case += 1
    save("G2_rgcn_layer", ncases=case, **out)


def gen_G4_G5():
    """GRRGCNLayer.forward (fixed & learnable decay, nn.GRU & type-1 cell) + aliasing check
    (models/RRGCN.py:64-89, models/GRU_cell.py:7-31)."""
    import dgl
    from models.RRGCN import GRRGCNLayer
    num_e, num_r, tr, va, te = graphs()
    times = list(tr.keys())
    gl = [tr[times[i]] for i in (5, 6)]
    sizes = [g.number_of_nodes() for g in gl]
    out = {}
    case = 0
    for (D, B, type1, learn, nl) in [(32, 16, False, False, 1), (32, 16, False, True, 1), (32, 16, True, False, 1),
                                     (200, 100, False, False, 1), (16, 4, False, False, 2)]:
        seed = 200 + case
        rng = np.random.default_rng(seed)
        args = rh.make_args(type1=type1, learnable_lambda=learn, num_layers=nl, inv_temperature=0.1)
        layer = GRRGCNLayer(args, D, D, 2 * num_r, B, times, bias=False, activation=None, self_loop=True, dropout=0.0)
        p = _layer_params(rng, D, B, 2 * num_r, len(times), False)
        _set_layer(layer, p)
        if type1:
            k = 1.0
            rp = [dict(w_ih=torch.from_numpy(rng.standard_normal((D, D)).astype(np.float32) * 0.1),
                       w_hh=torch.from_numpy(rng.standard_normal((3 * D, D)).astype(np.float32) * 0.1),
                       b_ih=torch.from_numpy(rng.standard_normal(D).astype(np.float32) * 0.1),
                       b_hh=torch.from_numpy(rng.standard_normal(3 * D).astype(np.float32) * 0.1))]
            layer.rnn.weight_ih.data.copy_(rp[0]['w_ih'])
            layer.rnn.weight_hh.data.copy_(rp[0]['w_hh'])
            layer.rnn.bias_ih.data.copy_(rp[0]['b_ih'])
            layer.rnn.bias_hh.data.copy_(rp[0]['b_hh'])
        else:
            rp = O._gru_params(rng, D, nl)
            for li, q in enumerate(rp):
                getattr(layer.rnn, 'weight_ih_l%d' % li).data.copy_(q['w_ih'])
                getattr(layer.rnn, 'weight_hh_l%d' % li).data.copy_(q['w_hh'])
                getattr(layer.rnn, 'bias_ih_l%d' % li).data.copy_(q['b_ih'])
                getattr(layer.rnn, 'bias_hh_l%d' % li).data.copy_(q['b_hh'])
        if learn:
            layer.exponential_decay.weight.data.fill_(0.3)
            layer.exponential_decay.bias.data.fill_(-0.2)
        ent = O._xavier(rng, num_e, D)
        bg = dgl.batch(gl)
        n = bg.number_of_nodes()
        h0 = ent[bg.ndata['id'].view(-1)].clone().requires_grad_(True)
        prev = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32) * 0.3)
        prev[rng.random(n) < 0.4] = 0
        prev = prev.requires_grad_(True)
        dt = torch.from_numpy(rng.integers(0, 6, (n, 1)).astype(np.float32))
        bg.ndata['h'] = h0
        g_ret, temb = layer(bg, prev, dt, [5, 6], sizes)
        aliased = int(g_ret is bg)
        hid = bg.ndata['h']
        gy = torch.from_numpy(np.random.default_rng(seed + 1000).standard_normal(tuple(hid.shape)).astype(np.float32))
        hid.backward(gy)
        pre = 'c%d_' % case
        if case == 0:
            out.update(graph_arrays(bg))
            out['node_sizes'] = sizes
        grads = {}
        if type1:
            grads = {'d_w_ih': layer.rnn.weight_ih.grad, 'd_w_hh': layer.rnn.weight_hh.grad,
                     'd_b_ih': layer.rnn.bias_ih.grad, 'd_b_hh': layer.rnn.bias_hh.grad}
        else:
            grads = {'d_w_ih': layer.rnn.weight_ih_l0.grad, 'd_w_hh': layer.rnn.weight_hh_l0.grad,
                     'd_b_ih': layer.rnn.bias_ih_l0.grad, 'd_b_hh': layer.rnn.bias_hh_l0.grad}
        out.update({pre + 'D': D, pre + 'B': B, pre + 'type1': int(type1), pre + 'learn': int(learn), pre + 'nl': nl,
                    pre + 'seed': seed, pre + 'prev': prev, pre + 'dt': dt.view(-1), pre + 'hid': hid,
                    pre + 'aliased': aliased, pre + 'd_h0': h0.grad, pre + 'd_prev': prev.grad,
                    pre + 'd_loop': layer.loop_weight.grad,
                    pre + 'd_weight_abs': layer.weight.grad.double().abs().sum().item()})
        if type1:
            for k2, v in rp[0].items():
                out[pre + 'rnn_' + k2] = v
        if learn:
            out[pre + 'd_decay_w'] = layer.exponential_decay.weight.grad
            out[pre + 'd_decay_b'] = layer.exponential_decay.bias.grad
        for k2, v in grads.items():
            out[pre + k2] = v
        case += 1
    save("G4_grrgcn_layer", ncases=case, **out)


def _encoder_case(enc_cls, module, rec_only, te, D, B, seed, bi):
    """Shared G6/G7/G8 driver: builds the reference container, loads seeded params, returns
    (encoder, oracle-format model, cfg)."""
    num_e, num_r, tr, va, te_g = graphs()
    times = np.array(list(tr.keys()))
    args = rh.make_args(module=module, rec_only_last_layer=rec_only, use_time_embedding=te, hidden_size=D,
                        embed_size=D, n_bases=B)
    cfg = dict(module=module, n_bases=B, inv_temperature=0.1, rec_only_last_layer=rec_only, use_time_embedding=te)
    model = O.init_model(cfg, num_e, num_r, len(times), D, seed=seed)
    enc = enc_cls(args, D, D, num_r, times)
    sd = {k[len('ent_encoder.'):]: v for k, v in to_ref_state_dict(model).items() if k.startswith('ent_encoder.')}
    enc.load_state_dict(sd, strict=True)
    return enc, model, cfg, args


def gen_G6():
    """RRGCN.forward / forward_isolated / forward_post_ensemble (models/RRGCN.py:170-253)."""
    import dgl
    from models.RRGCN import RRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    gl = [tr[times[i]] for i in (7, 8, 9)]
    sizes = [g.number_of_nodes() for g in gl]
    tl = [7, 8, 9]
    out = {}
    case = 0
    for (module, rec_only, te, D, B) in [('GRRGCN', True, False, 32, 16), ('GRRGCN', False, False, 32, 16),
                                         ('GRRGCN', True, True, 32, 16), ('GRRGCN', False, True, 16, 4),
                                         ('RRGCN', False, False, 32, 16), ('RRGCN', True, True, 32, 32),
                                         ('GRRGCN', True, False, 200, 100)]:
        seed = 300 + case
        enc, model, cfg, args = _encoder_case(RRGCN, module, rec_only, te, D, B, seed, False)
        rng = np.random.default_rng(seed + 5000)
        bg = dgl.batch(gl)
        n = bg.number_of_nodes()
        ent = model['ent_embeds'].clone().requires_grad_(True)
        bg.ndata['h'] = ent[bg.ndata['id'].view(-1)]
        p1 = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32) * 0.3).requires_grad_(True)
        p2 = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32) * 0.3).requires_grad_(True)
        dt = torch.from_numpy(rng.integers(0, 6, (n, 1)).astype(np.float32))
        first, second = enc(bg, p1, p2, dt, tl, sizes)
        gy = torch.from_numpy(np.random.default_rng(seed + 1000).standard_normal(tuple(second.shape)).astype(np.float32))
        (second * gy).sum().backward()
        pre = 'c%d_' % case
        if case == 0:
            out.update(graph_arrays(bg))
            out['node_sizes'] = sizes
            out['times'] = tl
        gsum = {k: v.grad.double().abs().sum().item() for k, v in enc.named_parameters() if v.grad is not None}
        out.update({pre + 'module': module, pre + 'rec_only': int(rec_only), pre + 'te': int(te), pre + 'D': D, pre + 'B': B,
                    pre + 'seed': seed, pre + 'second': second, pre + 'same': int(first is second),
                    pre + 'd_p1': (p1.grad if p1.grad is not None else torch.zeros_like(p1)), pre + 'd_p2': p2.grad,
                    pre + 'd_ent_rows': ent.grad[bg.ndata['id'].view(-1)],
                    pre + 'param_checksum': checksum(model)})
        if first is not second:
            out[pre + 'first'] = first
        for k, v in gsum.items():
            out[pre + 'gabs_' + k] = v
        # isolated pass over a slab of entities (models/RRGCN.py:206-217)
        with torch.no_grad():
            ne = 256
            e = model['ent_embeds'][:ne]
            q1 = torch.from_numpy(rng.standard_normal((ne, D)).astype(np.float32) * 0.3)
            q2 = torch.from_numpy(rng.standard_normal((ne, D)).astype(np.float32) * 0.3)
            dti = torch.from_numpy(rng.integers(0, 6, (ne, 1)).astype(np.float32))
            iso = enc.forward_isolated(e, q1, q2, dti, 8)
            out.update({pre + 'iso_q1': q1, pre + 'iso_q2': q2, pre + 'iso_dt': dti.view(-1), pre + 'iso': iso})
        # post-ensemble entry point (GRU module only; models/RRGCN.py:219-233)
        if module == 'GRRGCN':
            enc.layer_2.post_ensemble = True
            if not rec_only:
                enc.layer_1.post_ensemble = True
            with torch.no_grad():
                bg2 = dgl.batch(gl)
                bg2.ndata['h'] = model['ent_embeds'][bg2.ndata['id'].view(-1)]
                loc, f2, s2 = enc.forward_post_ensemble(bg2, p1.detach(), p2.detach(), dt, tl, sizes)
                out.update({pre + 'post_loc': loc, pre + 'post_second': s2})
        case += 1
    save("G6_rrgcn", ncases=case, **out)


def gen_G7():
    """BiRRGCN.forward / forward_one_direction / forward_isolated / post-ensemble
    (models/BiRRGCN.py:188-293)."""
    import dgl
    from models.BiRRGCN import BiRRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    gl = [tr[times[i]] for i in (10, 11)]
    sizes = [g.number_of_nodes() for g in gl]
    tl = [10, 11]
    out = {}
    case = 0
    for (module, rec_only, te, D, B) in [('BiGRRGCN', True, False, 32, 16), ('BiGRRGCN', False, False, 32, 16),
                                         ('BiGRRGCN', True, True, 16, 4), ('BiRRGCN', False, False, 32, 16),
                                         ('BiGRRGCN', True, False, 200, 100)]:
        seed = 400 + case
        enc, model, cfg, args = _encoder_case(BiRRGCN, module, rec_only, te, D, B, seed, True)
        rng = np.random.default_rng(seed + 5000)
        bg = dgl.batch(gl)
        n = bg.number_of_nodes()
        mk = lambda: torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32) * 0.3).requires_grad_(True)
        f1, f2, b1, b2 = mk(), mk(), mk(), mk()
        dtf = torch.from_numpy(rng.integers(0, 6, (n, 1)).astype(np.float32))
        dtb = torch.from_numpy(rng.integers(0, 6, (n, 1)).astype(np.float32))
        ent = model['ent_embeds'].clone().requires_grad_(True)
        bg.ndata['h'] = ent[bg.ndata['id'].view(-1)]
        second = enc(bg, f1, f2, dtf, b1, b2, dtb, tl, sizes)
        gy = torch.from_numpy(np.random.default_rng(seed + 1000).standard_normal(tuple(second.shape)).astype(np.float32))
        (second * gy).sum().backward()
        pre = 'c%d_' % case
        if case == 0:
            out.update(graph_arrays(bg))
            out['node_sizes'] = sizes
            out['times'] = tl
        z = lambda p: p.grad if p.grad is not None else torch.zeros_like(p)
        out.update({pre + 'module': module, pre + 'rec_only': int(rec_only), pre + 'te': int(te), pre + 'D': D, pre + 'B': B,
                    pre + 'seed': seed, pre + 'second': second, pre + 'd_f1': z(f1), pre + 'd_f2': z(f2), pre + 'd_b1': z(b1),
                    pre + 'd_b2': z(b2), pre + 'd_ent_rows': ent.grad[bg.ndata['id'].view(-1)],
                    pre + 'param_checksum': checksum(model)})
        for k, v in enc.named_parameters():
            if v.grad is not None:
                out[pre + 'gabs_' + k] = v.grad.double().abs().sum().item()
        with torch.no_grad():
            for fwd in (True, False):
                bg2 = dgl.batch(gl)
                bg2.ndata['h'] = model['ent_embeds'][bg2.ndata['id'].view(-1)]
                a, b = enc.forward_one_direction(bg2, f1.detach(), f2.detach(), dtf, tl, sizes, fwd)
                out[pre + ('one_fwd' if fwd else 'one_bwd')] = b
                out[pre + ('one_same_fwd' if fwd else 'one_same_bwd')] = int(a is b)
                if a is not b:
                    out[pre + ('one_first_fwd' if fwd else 'one_first_bwd')] = a
            ne = 256
            e = model['ent_embeds'][:ne]
            q = lambda: torch.from_numpy(rng.standard_normal((ne, D)).astype(np.float32) * 0.3)
            q1, q2, q3, q4 = q(), q(), q(), q()
            d1 = torch.from_numpy(rng.integers(0, 6, (ne, 1)).astype(np.float32))
            d2 = torch.from_numpy(rng.integers(0, 6, (ne, 1)).astype(np.float32))
            iso = enc.forward_isolated(e, q1, q2, d1, q3, q4, d2, 10)
            out.update({pre + 'iso_f1': q1, pre + 'iso_f2': q2, pre + 'iso_b1': q3, pre + 'iso_b2': q4,
                        pre + 'iso_dtf': d1.view(-1), pre + 'iso_dtb': d2.view(-1), pre + 'iso': iso})
            if module == 'BiGRRGCN':
                enc.layer_2.post_ensemble = True
                if not rec_only:
                    enc.layer_1.post_ensemble = True
                bg3 = dgl.batch(gl)
                bg3.ndata['h'] = model['ent_embeds'][bg3.ndata['id'].view(-1)]
                loc, s2 = enc.forward_post_ensemble(bg3, f1.detach(), f2.detach(), dtf, b1.detach(), b2.detach(), dtb, tl, sizes)
                out.update({pre + 'post_loc': loc, pre + 'post_second': s2})
        case += 1
    save("G7_birrgcn", ncases=case, **out)


def gen_G9():
    """utils/scores.py: 3 scorers x 3 modes."""
    from utils import scores as S
    rng = np.random.default_rng(9)
    P, K, D = 7, 5, 16
    s = torch.from_numpy(rng.standard_normal((P, D)).astype(np.float32))
    r = torch.from_numpy(rng.standard_normal((P, D)).astype(np.float32))
    o = torch.from_numpy(rng.standard_normal((P, D)).astype(np.float32))
    cand = torch.from_numpy(rng.standard_normal((P, K, D)).astype(np.float32))
    out = dict(s=s, r=r, o=o, cand=cand)
    for name in ('distmult', 'complex', 'transE'):
        fn = getattr(S, name)
        out[name + '_single'] = fn(s, r, o)
        out[name + '_tail'] = fn(s, r, cand, mode='tail')
        out[name + '_head'] = fn(cand, r, o, mode='head')
    save("G9_scores", **out)


def _run_window(cls, module, rec_only, D, B, seed, t_list, L, bsz_args, neg, trace=False, te=False, ent_row_stride=1, extra_args=None,
                tweak=None, backward=True):
    num_e, num_r, tr, va, te_g = graphs()
    args = rh.make_args(module=module, rec_only_last_layer=rec_only, hidden_size=D, embed_size=D, n_bases=B,
                        train_seq_len=L, test_seq_len=L, batch_size=bsz_args, negative_rate=neg, use_time_embedding=te, **(extra_args or {}))
    cfg = dict(module=module, n_bases=B, inv_temperature=0.1, rec_only_last_layer=rec_only, use_time_embedding=te)
    model = O.init_model(cfg, num_e, num_r, len(tr), D, seed=seed)
    torch.manual_seed(0)
    m = cls(args, num_e, num_r, tr, va, te_g)
    missing = m.load_state_dict(to_ref_state_dict(model), strict=False)
    assert not missing.unexpected_keys and all(("impute_weight" in k or "_linear" in k) for k in missing.missing_keys), missing
    if tweak is not None:
        tweak(m)
    np.random.seed(seed)
    # capture the reference's unseeded random draws (F11) -----------------------------
    choices, samples, trace_rec = [], [], []
    orig_choice = np.random.choice

    def rec_choice(*a, **k):
        r = orig_choice(*a, **k)
        choices.append(np.asarray(r).copy())
        return r

    orig_neg = m.corrupter.single_graph_negative_sampling

    def rec_neg(t, g, n):
        res = orig_neg(t, g, n)
        samples.append([x.clone() for x in res[:3]])
        return res

    m.corrupter.single_graph_negative_sampling = rec_neg
    if trace:
        orig_upd = m.update_time_diff_hist_embeddings

        def rec_upd(f, s, start, gl, cur_t, bsz):
            res = orig_upd(f, s, start, gl, cur_t, bsz)
            trace_rec.append((cur_t, res.detach().clone(), start.detach().clone()))
            return res

        m.update_time_diff_hist_embeddings = rec_upd
    np.random.choice = rec_choice
    try:
        loss = m(torch.tensor(t_list))
    finally:
        np.random.choice = orig_choice
    times = list(tr.keys())
    out = dict(module=module, rec_only=int(rec_only), D=D, B=B, seed=seed, L=L, neg=neg, te=int(te),
               t_list=np.array(t_list), times=np.array(times), loss=loss.item(), param_checksum=checksum(model),
               n_choices=len(choices), n_samples=len(samples))
    for i, c in enumerate(choices):
        out['choice_%d' % i] = c
    for i, (trip, nt, nh) in enumerate(samples):
        out['trip_%d' % i], out['negtail_%d' % i], out['neghead_%d' % i] = trip, nt, nh
    if not backward:
        return out
    loss.backward()
    eg = m.ent_embeds.grad
    nz = torch.nonzero(eg.abs().sum(1)).view(-1)
    if ent_row_stride > 1:                 # large-D fixtures: every k-th non-zero row + the global sums (gabs_/gsum_ent_embeds)
        nz = nz[::ent_row_stride]
        out['d_ent_sub'] = ent_row_stride
    out['d_ent_nz_rows'] = nz
    out['d_ent_nz_vals'] = eg[nz]
    out['d_rel'] = m.rel_embeds.grad
    for k, v in m.named_parameters():
        if v.grad is not None:
            out['gabs_' + k] = v.grad.double().abs().sum().item()
            out['gsum_' + k] = v.grad.double().sum().item()
    if trace:
        out['n_trace'] = len(trace_rec)
        for i, (cur_t, res, start) in enumerate(trace_rec):
            out['tr%d_cur_t' % i] = cur_t
            for b in range(res.shape[0]):
                rows = torch.nonzero(res[b, 1].abs().sum(1)).view(-1)
                out['tr%d_b%d_rows' % (i, b)] = rows
                out['tr%d_b%d_vals' % (i, b)] = res[b, 1][rows]
                out['tr%d_b%d_same' % (i, b)] = int(torch.equal(res[b, 0], res[b, 1]))
                srows = torch.nonzero(start[b]).view(-1)
                out['tr%d_b%d_srows' % (i, b)] = srows
                out['tr%d_b%d_svals' % (i, b)] = start[b][srows]
    return out


def gen_G10():
    """Window level: DynamicRGCN.forward / BiDynamicRGCN.forward loss + grads on ICEWS14
    (models/DynamicRGCN.py:176-194, models/BiDynamicRGCN.py:123-144); t=3 exercises the
    None-padded window, t=0 the all-padding extreme.  G11 = the history trace (F8)."""
    from models.DynamicRGCN import DynamicRGCN
    from models.BiDynamicRGCN import BiDynamicRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    T = lambda idx: [int(times[i]) for i in idx]
    save("G10_uni_grrgcn", **_run_window(DynamicRGCN, 'GRRGCN', False, 32, 16, 501, T([20, 15, 9, 3]), 8, 4, 20, trace=True))
    save("G10_uni_grrgcn_rol", **_run_window(DynamicRGCN, 'GRRGCN', True, 32, 16, 502, T([12, 5, 0]), 6, 4, 20))
    save("G10_bi_grrgcn_rol", **_run_window(BiDynamicRGCN, 'BiGRRGCN', True, 32, 16, 503, T([20, 15, 9, 3]), 8, 4, 20))
    save("G10_bi_grrgcn", **_run_window(BiDynamicRGCN, 'BiGRRGCN', False, 16, 4, 504, T([22, 10, 1]), 5, 4, 20))
    save("G10_uni_grrgcn_d200", **_run_window(DynamicRGCN, 'GRRGCN', True, 200, 100, 505, T([9, 4]), 4, 4, 10))
    # BASELINE's headline shape at window level: bidirectional, L=15, D=200 / 100 bases, rec-only-last-layer (config 4's model on
    # the ICEWS14 slice; windows are clipped by the slice's 24 timestamps exactly as the reference clips them at the data's ends)
    save("G10_bi_grrgcn_rol_d200", **_run_window(BiDynamicRGCN, 'BiGRRGCN', True, 200, 100, 506, T([21, 14, 8, 2]), 15, 4, 10,
                                                 ent_row_stride=9))


def gen_G12():
    """StaticRGCN (config 1): baselines/StaticRGCN.py:36-89, models/RGCN.py:145-164."""
    from baselines.StaticRGCN import StaticRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    D, B, seed = 32, 16, 601
    args = rh.make_args(module='SRGCN', hidden_size=D, embed_size=D, n_bases=B, negative_rate=20)
    cfg = dict(module='SRGCN', n_bases=B, inv_temperature=0.1, rec_only_last_layer=False, use_time_embedding=False)
    model = O.init_model(cfg, num_e, num_r, len(tr), D, seed=seed, bias=True)
    rng = np.random.default_rng(seed + 1)
    for ln in ('layer_1', 'layer_2'):
        model['ent_encoder'][ln]['h_bias'] = torch.from_numpy(rng.uniform(-0.3, 0.3, D).astype(np.float32))
    m = StaticRGCN(args, num_e, num_r, tr, va, te_g)
    m.load_state_dict(to_ref_state_dict(model), strict=True)
    t_list = torch.tensor([int(times[i]) for i in (2, 6)])
    gl = [tr[t.item()] for t in t_list]
    with torch.no_grad():
        embeds = m.get_per_graph_ent_embeds(t_list, gl, val=True)
        iso = m.ent_encoder.forward_isolated(m.ent_embeds[:200], t_list[0])
    out = dict(D=D, B=B, seed=seed, t_list=t_list.numpy(), times=np.array(times), iso=iso, param_checksum=checksum(model), neg=20)
    for i, e in enumerate(embeds):
        out['emb_%d' % i] = e
    # training step (baselines/StaticRGCN.py:36-46): loss + gradients with the reference's random draws recorded
    np.random.seed(seed)
    choices, samples = [], []
    orig_choice = np.random.choice

    def rec_choice(*a, **k):
        r = orig_choice(*a, **k)
        choices.append(np.asarray(r).copy())
        return r

    orig_neg = m.corrupter.single_graph_negative_sampling

    def rec_neg(t, g, n):
        res = orig_neg(t, g, n)
        samples.append([x.clone() for x in res[:3]])
        return res

    m.corrupter.single_graph_negative_sampling = rec_neg
    np.random.choice = rec_choice
    try:
        loss = m(t_list)
    finally:
        np.random.choice = orig_choice
    loss.backward()
    out.update(loss=loss.item(), n_choices=len(choices), n_samples=len(samples))
    for i, c in enumerate(choices):
        out['choice_%d' % i] = c
    for i, (trip, nt, nh) in enumerate(samples):
        out['trip_%d' % i], out['negtail_%d' % i], out['neghead_%d' % i] = trip, nt, nh
    eg = m.ent_embeds.grad
    nz = torch.nonzero(eg.abs().sum(1)).view(-1)
    out['d_ent_nz_rows'], out['d_ent_nz_vals'], out['d_rel'] = nz, eg[nz], m.rel_embeds.grad
    for k, v in m.named_parameters():
        if v.grad is not None:
            out['gabs_' + k] = v.grad.double().abs().sum().item()
    for ln in ('layer_1', 'layer_2'):
        out['d_bias_' + ln] = getattr(m.ent_encoder, ln).h_bias.grad
    save("G12_static_rgcn", **out)


G13_REL_SCALE = 250.0      # rel_embeds multiplier: the untrained D=32 model scores every candidate within 1e-3 of 0, i.e.
                            # sigmoid = 0.5 +- 2e-4 in fp32 -- hundreds of exact ties per row; scaled, the scores spread over ~[-3, 3]
G13_BAND = 1.5e-6           # |sigmoid(score_e) - sigmoid(score_target)| below this is "inside the tie band" (fp32 rounding of
                            # two different but equally valid evaluation orders can swap such a pair)


def gen_G13():
    """evaluate(): filtered ranks + classification loss (models/DynamicRGCN.py:118-144,196-220,
    models/BiDynamicRGCN.py:146-208, utils/evaluation.py:34-106).  For every ranked row the generator also records
    `nclose` = the number of unfiltered competitors whose sigmoid score lies within G13_BAND of the target's: rows with
    nclose == 0 have a rank that no rounding can move (tests demand exact equality there), the others may move by at most
    nclose."""
    from models.DynamicRGCN import DynamicRGCN
    from models.BiDynamicRGCN import BiDynamicRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    for name, cls, module, rec_only, seed, idx in (("G13_eval_uni", DynamicRGCN, 'GRRGCN', True, 701, [14, 8, 2]),
                                                  ("G13_eval_bi", BiDynamicRGCN, 'BiGRRGCN', True, 702, [21, 12, 6])):
        D, B, L = 32, 16, 6
        args = rh.make_args(module=module, rec_only_last_layer=rec_only, hidden_size=D, embed_size=D, n_bases=B,
                            train_seq_len=L, test_seq_len=L, batch_size=4, negative_rate=20)
        cfg = dict(module=module, n_bases=B, inv_temperature=0.1, rec_only_last_layer=rec_only, use_time_embedding=False)
        model = O.init_model(cfg, num_e, num_r, len(tr), D, seed=seed)
        model['rel_embeds'] = model['rel_embeds'] * G13_REL_SCALE
        m = cls(args, num_e, num_r, tr, va, te_g)
        m.load_state_dict(to_ref_state_dict(model), strict=True)
        t_list = [int(times[i]) for i in idx]
        out = dict(module=module, rec_only=int(rec_only), D=D, B=B, seed=seed, L=L, te=0, neg=20, t_list=np.array(t_list),
                   param_checksum=checksum(model), rel_scale=G13_REL_SCALE, band=G13_BAND)
        ev = m.evaluater
        rec = dict(mode=None, graphs=[])
        orig_single, orig_perturb, orig_sort = ev.calc_metrics_single_graph, ev.perturb_and_get_rank, ev.sort_and_rank

        def single(*a, **k):
            rec['graphs'].append(dict(head=[], tail=[]))
            return orig_single(*a, **k)

        def perturb(*a, **k):
            rec['mode'] = k.get('mode', a[-1] if a and isinstance(a[-1], str) else 'tail')
            return orig_perturb(*a, **k)

        def sort_and_rank(score, target):
            ts = score.gather(1, target.view(-1, 1))
            d = (score - ts).abs()
            d.scatter_(1, target.view(-1, 1), float('inf'))
            rec['graphs'][-1][rec['mode']].append(((d <= G13_BAND) & (score > 1e-30)).sum(1))
            rec.setdefault('spread', []).append(score[score > 1e-30].std().item())
            return orig_sort(score, target)

        ev.calc_metrics_single_graph, ev.perturb_and_get_rank, ev.sort_and_rank = single, perturb, sort_and_rank
        with torch.no_grad():
            for split, val in (("val", True), ("test", False)):
                rec['graphs'] = []
                ranks, loss = m.evaluate(torch.tensor(t_list), val=val)
                nclose = torch.cat([torch.cat(g['head'] + g['tail']) for g in rec['graphs']])     # ranks = [head ranks ; tail ranks] per graph
                assert nclose.shape == ranks.shape
                out["ranks_" + split] = ranks
                out["nclose_" + split] = nclose
                out["loss_" + split] = float(loss)
                print("  %s %s: %d ranks, %.1f%% outside every tie band, sigmoid spread %.3f, mean rank %.1f" %
                      (name, split, ranks.numel(), 100.0 * (nclose == 0).float().mean().item(), float(np.mean(rec['spread'])),
                       ranks.float().mean().item()))
        save(name, **out)


def gen_G14():
    """Self-attention models (config 5): SelfAttentionRGCN / BiSelfAttentionRGCN forward loss + grads
    (models/SelfAttentionRGCN.py:122-140, models/BiSelfAttentionRGCN.py:48-69, models/SARGCN.py)."""
    from models.SelfAttentionRGCN import SelfAttentionRGCN
    from models.BiSelfAttentionRGCN import BiSelfAttentionRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    T = lambda idx: [int(times[i]) for i in idx]
    for name, cls, module, rec_only, learn, seed, idx, L in (
            ("G14_sa_uni_rol", SelfAttentionRGCN, 'SARGCN', True, False, 801, [18, 11, 3], 6),
            ("G14_sa_uni", SelfAttentionRGCN, 'SARGCN', False, True, 802, [20, 9, 1], 5),
            ("G14_sa_bi_rol", BiSelfAttentionRGCN, 'BiSARGCN', True, False, 803, [19, 12, 4], 5)):
        out = _run_window_sa(cls, module, rec_only, learn, 32, 16, seed, T(idx), L, 20)
        save(name, **out)


def _run_window_sa(cls, module, rec_only, learn, D, B, seed, t_list, L, neg):
    num_e, num_r, tr, va, te_g = graphs()
    args = rh.make_args(module=module, rec_only_last_layer=rec_only, hidden_size=D, embed_size=D, n_bases=B, train_seq_len=L,
                        test_seq_len=L, batch_size=4, negative_rate=neg, learnable_lambda=learn, EMA=False)
    cfg = dict(module=module, n_bases=B, inv_temperature=0.1, rec_only_last_layer=rec_only, use_time_embedding=True, learnable_lambda=learn)
    model = O.init_model(cfg, num_e, num_r, len(tr), D, seed=seed)
    if learn:
        for ln in ('layer_1', 'layer_2'):
            model['ent_encoder'][ln]['exponential_decay'] = (torch.full((1, 1), 0.25), torch.full((1,), -0.1))
    torch.manual_seed(0)
    m = cls(args, num_e, num_r, tr, va, te_g)
    missing = m.load_state_dict(to_ref_state_dict(model), strict=False)
    assert not missing.unexpected_keys, missing
    assert all('exponential_decay' in k for k in missing.missing_keys) or not missing.missing_keys, missing
    np.random.seed(seed)
    choices, samples = [], []
    orig_choice = np.random.choice

    def rec_choice(*a, **k):
        r = orig_choice(*a, **k)
        choices.append(np.asarray(r).copy())
        return r

    orig_neg = m.corrupter.single_graph_negative_sampling

    def rec_neg(t, g, n):
        res = orig_neg(t, g, n)
        samples.append([x.clone() for x in res[:3]])
        return res

    m.corrupter.single_graph_negative_sampling = rec_neg
    np.random.choice = rec_choice
    try:
        loss = m(torch.tensor(t_list))
    finally:
        np.random.choice = orig_choice
    loss.backward()
    times = list(tr.keys())
    out = dict(module=module, rec_only=int(rec_only), learn=int(learn), D=D, B=B, seed=seed, L=L, neg=neg, te=1,
               t_list=np.array(t_list), times=np.array(times), loss=loss.item(), param_checksum=checksum(model),
               n_choices=len(choices), n_samples=len(samples))
    for i, c in enumerate(choices):
        out['choice_%d' % i] = c
    for i, (trip, nt, nh) in enumerate(samples):
        out['trip_%d' % i], out['negtail_%d' % i], out['neghead_%d' % i] = trip, nt, nh
    eg = m.ent_embeds.grad
    nz = torch.nonzero(eg.abs().sum(1)).view(-1)
    out['d_ent_nz_rows'] = nz
    out['d_ent_nz_vals'] = eg[nz]
    out['d_rel'] = m.rel_embeds.grad
    for k, v in m.named_parameters():
        if v.grad is not None:
            out['gabs_' + k] = v.grad.double().abs().sum().item()
    return out


def _sparse_rows(prefix, t, out):
    """dense (N, D) tensor -> non-zero rows + values under `prefix`."""
    nz = torch.nonzero(t.abs().sum(1)).view(-1)
    out[prefix + "_rows"], out[prefix + "_vals"] = nz, t[nz]


def gen_G15():
    """Config 3 at window level: the post-ensemble / impute window loops (models/PostBiDynamicRGCN.py:77-101,103-124,
    models/PostDynamicRGCN.py:29-96) with the three history streams (local, layer-1, layer-2) and the entry points they call:
    BiRRGCN.forward_post_ensemble_one_direction / forward_post_ensemble / forward_post_ensemble_isolated /
    forward_isolated_impute (models/BiRRGCN.py:259-338) and their RRGCN counterparts (models/RRGCN.py:219-272).
      G15_post_bi    BiGRRGCN --rec-only-last-layer --post-ensemble, L = 15: (local, temporal) target embeddings, the local
                     history streams, the all-entity (local, temporal) matrices of every window, gradients of a seeded
                     weighted sum of all of them.  (The frequency-gated score combination of PostEnsembleBiDynamicRGCN is
                     out of scope)
      G15_impute_bi  ImputeBiDynamicRGCN.forward (--impute): loss + gradients with the recorded draws.
      G15_impute_uni ImputeDynamicRGCN.forward."""
    from models.PostBiDynamicRGCN import ImputeBiDynamicRGCN
    from models.PostDynamicRGCN import ImputeDynamicRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    T = lambda idx: [int(times[i]) for i in idx]

    # ---- post-ensemble, bidirectional, rec-only-last-layer (BASELINE config 3's flags) ---------------------------------------
    D, B, L, seed = 32, 16, 15, 901
    t_list = T([21, 13, 7, 2])
    args = rh.make_args(module='BiGRRGCN', rec_only_last_layer=True, post_ensemble=True, hidden_size=D, embed_size=D, n_bases=B,
                        train_seq_len=L, test_seq_len=L, batch_size=4, negative_rate=20)
    cfg = dict(module='BiGRRGCN', n_bases=B, inv_temperature=0.1, rec_only_last_layer=True, use_time_embedding=False)
    model = O.init_model(cfg, num_e, num_r, len(tr), D, seed=seed)
    m = ImputeBiDynamicRGCN(args, num_e, num_r, tr, va, te_g)
    m.load_state_dict(to_ref_state_dict(model), strict=True)
    np.random.seed(seed)
    choices = []
    orig_choice = np.random.choice

    def rec_choice(*a, **k):
        r = orig_choice(*a, **k)
        choices.append(np.asarray(r).copy())
        return r

    np.random.choice = rec_choice
    try:
        gf, tf, gb, tb = m.get_batch_graph_list(torch.tensor(t_list), L, m.graph_dict_train)
        f_loc, f_rec, f_start = m.pre_forward(gf, tf, forward=True)
        b_loc, b_rec, b_start = m.pre_forward(gb, tb, forward=False)
        train_graphs, tt = gf[-1], tf[-1]
        loc, rec = m.get_final_graph_embeds(train_graphs, tt, L, f_loc, f_rec, f_start, b_loc, b_rec, b_start, full=False)
    finally:
        np.random.choice = orig_choice
    out = dict(module='BiGRRGCN', rec_only=1, D=D, B=B, seed=seed, L=L, neg=20, te=0, t_list=np.array(t_list), times=np.array(times),
               param_checksum=checksum(model), n_choices=len(choices), bsz=len(loc))
    for i, c in enumerate(choices):
        out['choice_%d' % i] = c
    gen = torch.Generator().manual_seed(seed)
    total = 0
    rows_sel = torch.arange(0, num_e, 7)
    for i, (g, t) in enumerate(zip(train_graphs, tt)):
        out['loc_%d' % i], out['rec_%d' % i] = loc[i], rec[i]
        _sparse_rows('f_loc_%d' % i, f_loc[i], out)
        _sparse_rows('b_loc_%d' % i, b_loc[i], out)
        dtf = (L - 1 - f_start[i]).unsqueeze(-1)
        dtb = (L - 1 - b_start[i]).unsqueeze(-1)
        a_loc, a_rec = m.ent_encoder.forward_post_ensemble_isolated(m.ent_embeds, f_rec[i][0], f_rec[i][1], dtf, b_rec[i][0], b_rec[i][1],
                                                                     dtb, t, f_loc[i], b_loc[i])
        out['all_loc_%d' % i], out['all_rec_%d' % i] = a_loc[rows_sel], a_rec[rows_sel]
        for x in (loc[i], rec[i], a_loc[rows_sel], a_rec[rows_sel]):
            total = total + (x * torch.randn(x.shape, generator=gen)).sum()
    out['all_rows'] = rows_sel
    out['total'] = total.item()
    total.backward()
    eg = m.ent_embeds.grad
    nz = torch.nonzero(eg.abs().sum(1)).view(-1)[::5]
    out['d_ent_sub'], out['d_ent_nz_rows'], out['d_ent_nz_vals'] = 5, nz, eg[nz]
    for k, v in m.named_parameters():
        if v.grad is not None:
            out['gabs_' + k] = v.grad.double().abs().sum().item()
            out['gsum_' + k] = v.grad.double().sum().item()
    save("G15_post_bi", **out)

    # ---- impute models: the reference's own forward() ---------------------------------------------------------------------------
    for name, cls, module, rec_only, seed, idx, L in (("G15_impute_bi", ImputeBiDynamicRGCN, 'BiGRRGCN', True, 902, [20, 12, 3], 6),
                                                     ("G15_impute_uni", ImputeDynamicRGCN, 'GRRGCN', True, 903, [17, 9, 2], 6),
                                                     ("G15_impute_uni_full", ImputeDynamicRGCN, 'GRRGCN', False, 904, [15, 6], 5)):
        out = _run_window(cls, module, rec_only, 32, 16, seed, T(idx), L, 4, 20, extra_args=dict(impute=True),
                          tweak=_impute_weights)
        out['impute'] = 1
        save(name, **out)


def _impute_weights(m):
    """nn.Linear(1, 1) impute gates: set to fixed non-trivial values (the reference leaves them at torch's default init)."""
    enc = m.ent_encoder
    with torch.no_grad():
        for nm, (w, b) in (("impute_weight", (0.3, -0.1)), ("impute_weight_forward", (0.25, -0.05)), ("impute_weight_backward", (0.4, 0.1))):
            if hasattr(enc, nm):
                getattr(enc, nm).weight.fill_(w)
                getattr(enc, nm).bias.fill_(b)


def gen_G16():
    """evaluate() of the impute models (models/PostDynamicRGCN.py:101-143, models/PostBiDynamicRGCN.py:126-176): the window loop
    with the local history stream on the full train graphs, the imputed all-entity matrix (forward_isolated_impute), then the
    standard filtered ranks (utils/evaluation.py) and classification loss.  Tie bands recorded as for G13."""
    from models.PostBiDynamicRGCN import ImputeBiDynamicRGCN
    from models.PostDynamicRGCN import ImputeDynamicRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    for name, cls, module, seed, idx in (("G16_eval_impute_uni", ImputeDynamicRGCN, 'GRRGCN', 711, [14, 8, 2]),
                                         ("G16_eval_impute_bi", ImputeBiDynamicRGCN, 'BiGRRGCN', 712, [21, 12, 6])):
        D, B, L = 32, 16, 6
        args = rh.make_args(module=module, rec_only_last_layer=True, hidden_size=D, embed_size=D, n_bases=B,
                            train_seq_len=L, test_seq_len=L, batch_size=4, negative_rate=20, impute=True)
        cfg = dict(module=module, n_bases=B, inv_temperature=0.1, rec_only_last_layer=True, use_time_embedding=False)
        model = O.init_model(cfg, num_e, num_r, len(tr), D, seed=seed)
        csum = checksum(model)
        model['rel_embeds'] = model['rel_embeds'] * G13_REL_SCALE
        m = cls(args, num_e, num_r, tr, va, te_g)
        missing = m.load_state_dict(to_ref_state_dict(model), strict=False)
        assert not missing.unexpected_keys and all("impute_weight" in k for k in missing.missing_keys), missing
        _impute_weights(m)
        t_list = [int(times[i]) for i in idx]
        out = dict(module=module, rec_only=1, D=D, B=B, seed=seed, L=L, te=0, neg=20, t_list=np.array(t_list), impute=1,
                   param_checksum=csum, rel_scale=G13_REL_SCALE, band=G13_BAND)
        ev = m.evaluater
        rec = dict(mode=None, graphs=[])
        orig_single, orig_perturb, orig_sort = ev.calc_metrics_single_graph, ev.perturb_and_get_rank, ev.sort_and_rank

        def single(*a, **k):
            rec['graphs'].append(dict(head=[], tail=[]))
            return orig_single(*a, **k)

        def perturb(*a, **k):
            rec['mode'] = k.get('mode', a[-1] if a and isinstance(a[-1], str) else 'tail')
            return orig_perturb(*a, **k)

        def sort_and_rank(score, target):
            ts = score.gather(1, target.view(-1, 1))
            d = (score - ts).abs()
            d.scatter_(1, target.view(-1, 1), float('inf'))
            rec['graphs'][-1][rec['mode']].append(((d <= G13_BAND) & (score > 1e-30)).sum(1))
            return orig_sort(score, target)

        ev.calc_metrics_single_graph, ev.perturb_and_get_rank, ev.sort_and_rank = single, perturb, sort_and_rank
        with torch.no_grad():
            for split, val in (("val", True), ("test", False)):
                rec['graphs'] = []
                ranks, loss = m.evaluate(torch.tensor(t_list), val=val)
                nclose = torch.cat([torch.cat(g['head'] + g['tail']) for g in rec['graphs']])
                assert nclose.shape == ranks.shape
                out["ranks_" + split] = ranks
                out["nclose_" + split] = nclose
                out["loss_" + split] = float(loss)
                print("  %s %s: %d ranks, %.1f%% outside every tie band, mean rank %.1f" %
                      (name, split, ranks.numel(), 100.0 * (nclose == 0).float().mean().item(), ranks.float().mean().item()))
        save(name, **out)


def det_matrix(n, d, c):
    """Deterministic pseudo-embeddings shared by the generator and the tests (no storage): x[i, j] = 0.6 sin(0.37 i + 1.3 j + c)
    + 0.4 cos(0.011 i j + 2 c)."""
    i = torch.arange(n, dtype=torch.float64).view(-1, 1)
    j = torch.arange(d, dtype=torch.float64).view(1, -1)
    return (0.6 * torch.sin(0.37 * i + 1.3 * j + c) + 0.4 * torch.cos(0.011 * i * j + 2.0 * c)).float()


G17_REL_SCALE = float(os.environ.get('G17_REL_SCALE', '3.0'))


def gen_G17():
    """The post-ensemble evaluation filters (utils/post_evaluation.py): PostEvaluationFilter (embedding-level mix with four
    weights, lines 7-60) and PostEnsembleEvaluationFilter (score-level mix, lines 63-134), called directly with deterministic
    (local, temporal) embeddings and per-triple weights on the valid triples of one timestamp of the slice."""
    from models.DynamicRGCN import DynamicRGCN
    from utils.post_evaluation import PostEvaluationFilter, PostEnsembleEvaluationFilter
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    D = 16
    for sf in ("complex", "distmult"):
        args = rh.make_args(module='GRRGCN', rec_only_last_layer=True, hidden_size=D, embed_size=D, n_bases=8, train_seq_len=4,
                            test_seq_len=4, score_function=sf)
        m = DynamicRGCN(args, num_e, num_r, tr, va, te_g)
        t = int(times[int(os.environ.get('G17_T', '15'))])
        g = va[t]
        n_g = g.number_of_nodes()
        src, dst = g.edges()
        samples = torch.stack([src, g.edata['type_s'], dst]).transpose(0, 1)
        P = samples.shape[0]
        loc, rec = det_matrix(n_g, D, 0.1), det_matrix(n_g, D, 0.7)
        all_loc, all_rec = det_matrix(num_e, D, 1.3), det_matrix(num_e, D, 2.1)
        rel = det_matrix(2 * num_r, D, 3.3) * G17_REL_SCALE    # keeps the sigmoid scores spread and unsaturated
        w = [(0.15 + 0.7 * torch.rand(P, 1, generator=torch.Generator().manual_seed(50 + k))) for k in range(4)]
        out = dict(score_function=sf, D=D, t=t, P=P, band=G13_BAND, rel_scale=G17_REL_SCALE)
        for k in range(4):
            out["w%d" % k] = w[k]
        for cls_name, cls in (("post", PostEvaluationFilter), ("ens", PostEnsembleEvaluationFilter)):
            ev = cls(args, m.calc_score, tr, va, te_g)
            nclose = []
            orig_sort = ev.sort_and_rank

            def sort_and_rank(score, target):
                ts = score.gather(1, target.view(-1, 1))
                dd = (score - ts).abs()
                dd.scatter_(1, target.view(-1, 1), float('inf'))
                nclose.append(((dd <= G13_BAND) & (score > 1e-30)).sum(1))
                return orig_sort(score, target)

            ev.sort_and_rank = sort_and_rank
            with torch.no_grad():
                if cls_name == "post":
                    ranks = ev.calc_metrics_single_graph(loc, rec, rel, all_loc, all_rec, samples, w[0], w[1], w[2], w[3], g, torch.tensor(t))
                    nc = torch.cat(nclose)                   # recorded tail first, then head; ranks = [head ; tail]
                    nc = torch.cat([nc[P:], nc[:P]])
                else:
                    ranks = ev.calc_metrics_single_graph(loc, rec, rel, all_loc, all_rec, w[0], w[1], samples, g, torch.tensor(t))
                    nc = torch.cat(nclose)
                    nc = torch.cat([nc[P:], nc[:P]])
            assert nc.shape == ranks.shape
            out["ranks_" + cls_name], out["nclose_" + cls_name] = ranks, nc
            print("  G17_%s %s: %d ranks, %.1f%% outside every tie band, mean rank %.1f" %
                  (sf, cls_name, ranks.numel(), 100.0 * (nc == 0).float().mean().item(), ranks.float().mean().item()))
        save("G17_post_eval_" + sf, **out)


def g18_ratio(triples, t, g):
    """The deterministic stand-in for calc_ensemble_ratio used by G18 (the reference derives the weights from the frequency
    tables of utils/DropEdge.py, outside the encoder path): per-triple weights as a function of the row index."""
    i = torch.arange(triples.shape[0], dtype=torch.float32).view(-1, 1)
    return 0.2 + 0.6 * torch.sin(0.7 * i + 0.1) ** 2, 0.25 + 0.5 * torch.cos(0.3 * i) ** 2


def gen_G18():
    """evaluate() of the score-level post-ensemble models (models/PostDynamicRGCN.py:367-423, models/PostBiDynamicRGCN.py:297-360)
    with calc_ensemble_ratio replaced by g18_ratio: window loop with the local stream, (local, temporal) all-entity matrices,
    PostEnsembleEvaluationFilter."""
    from models.PostBiDynamicRGCN import PostEnsembleBiDynamicRGCN
    from models.PostDynamicRGCN import PostEnsembleDynamicRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    for name, cls, module, seed, idx in (("G18_eval_post_uni", PostEnsembleDynamicRGCN, 'GRRGCN', 721, [14, 8, 2]),
                                         ("G18_eval_post_bi", PostEnsembleBiDynamicRGCN, 'BiGRRGCN', 722, [21, 12, 6])):
        D, B, L = 32, 16, 6
        args = rh.make_args(module=module, rec_only_last_layer=True, hidden_size=D, embed_size=D, n_bases=B,
                            train_seq_len=L, test_seq_len=L, batch_size=4, negative_rate=20, post_ensemble=True)
        cfg = dict(module=module, n_bases=B, inv_temperature=0.1, rec_only_last_layer=True, use_time_embedding=False)
        model = O.init_model(cfg, num_e, num_r, len(tr), D, seed=seed)
        csum = checksum(model)
        model['rel_embeds'] = model['rel_embeds'] * G13_REL_SCALE
        m = cls(args, num_e, num_r, tr, va, te_g)
        missing = m.load_state_dict(to_ref_state_dict(model), strict=False)
        assert not missing.unexpected_keys and all("_linear" in k for k in missing.missing_keys), missing
        m.calc_ensemble_ratio = g18_ratio
        t_list = [int(times[i]) for i in idx]
        out = dict(module=module, rec_only=1, D=D, B=B, seed=seed, L=L, te=0, neg=20, t_list=np.array(t_list), post_ensemble=1,
                   param_checksum=csum, rel_scale=G13_REL_SCALE, band=G13_BAND)
        ev = m.evaluater
        rec = dict(graphs=[])
        orig_single, orig_sort = ev.calc_metrics_single_graph, ev.sort_and_rank

        def single(*a, **k):
            rec['graphs'].append([])
            return orig_single(*a, **k)

        def sort_and_rank(score, target):
            ts = score.gather(1, target.view(-1, 1))
            d = (score - ts).abs()
            d.scatter_(1, target.view(-1, 1), float('inf'))
            rec['graphs'][-1].append(((d <= G13_BAND) & (score > 1e-30)).sum(1))
            return orig_sort(score, target)

        ev.calc_metrics_single_graph, ev.sort_and_rank = single, sort_and_rank
        with torch.no_grad():
            for split, val in (("val", True), ("test", False)):
                rec['graphs'] = []
                ranks, _ = m.evaluate(torch.tensor(t_list), val=val)
                # per graph the filter ranks tails first, then heads; ranks = [head ranks ; tail ranks]
                nclose = torch.cat([torch.cat([g[1], g[0]]) for g in rec['graphs']])
                assert nclose.shape == ranks.shape
                out["ranks_" + split], out["nclose_" + split] = ranks, nclose
                print("  %s %s: %d ranks, %.1f%% outside every tie band, mean rank %.1f" %
                      (name, split, ranks.numel(), 100.0 * (nclose == 0).float().mean().item(), ranks.float().mean().item()))
        save(name, **out)


def gen_G19():
    """PostEnsemble(Bi)DynamicRGCN.forward with the reference's OWN calc_ensemble_ratio (models/PostDynamicRGCN.py:375-397,425-461,
    models/PostBiDynamicRGCN.py:329-354): frequency tables of utils/DropEdge.py:34-82 (built by the reference from its full
    ICEWS14 train.txt; the target timestamps lie far enough inside the committed slice that every window the tables aggregate over
    is inside it too), the two 3-3-1 MLPs with recorded weights, the loss with the recorded draws, and the raw feature rows the
    MLPs were fed."""
    from models.PostBiDynamicRGCN import PostEnsembleBiDynamicRGCN
    from models.PostDynamicRGCN import PostEnsembleDynamicRGCN
    num_e, num_r, tr, va, te_g = graphs()
    times = list(tr.keys())
    for name, cls, module, seed, idx in (("G19_post_ratio_uni", PostEnsembleDynamicRGCN, 'GRRGCN', 731, [14, 9, 20]),
                                         ("G19_post_ratio_bi", PostEnsembleBiDynamicRGCN, 'BiGRRGCN', 732, [12, 18, 7])):
        L = 6
        assert [int(x) for x in times] == list(range(len(times)))
        assert max(idx) + (L if module.startswith("Bi") else 0) <= len(times)      # the tables' windows stay inside the slice
        feats = dict(sub=[], obj=[])
        mlp = {}

        def tweak(m):
            rng = np.random.default_rng(seed + 5)
            for nm in ("subject_linear", "object_linear"):
                seq = getattr(m, nm)
                for k, p in seq.named_parameters():
                    v = torch.from_numpy(rng.uniform(-0.6, 0.6, tuple(p.shape)).astype(np.float32))
                    p.data.copy_(v)
                    mlp["mlp_%s.%s" % (nm, k)] = v.clone()
            m.subject_linear.register_forward_pre_hook(lambda mod, inp: feats["sub"].append(inp[0].detach().clone()))
            m.object_linear.register_forward_pre_hook(lambda mod, inp: feats["obj"].append(inp[0].detach().clone()))

        out = _run_window(cls, module, True, 32, 16, seed, [int(times[i]) for i in idx], L, 4, 20, extra_args=dict(post_ensemble=True), tweak=tweak,
                          backward=False)      # the reference's own backward of this forward() fails under torch 2.x (an in-place
                                               # row overwrite of the all-entity matrices after they were saved for backward)
        out.update(mlp)
        out["post_ensemble"] = 1
        assert len(feats["sub"]) == len(idx) and len(feats["obj"]) == len(idx)
        for i in range(len(idx)):
            out["feat_sub_%d" % i], out["feat_obj_%d" % i] = feats["sub"][i], feats["obj"][i]
        print("  %s: loss %.6f, features up to %.0f" % (name, out["loss"], max(float(f.max()) for f in feats["sub"] + feats["obj"])))
        save(name, **out)


ALL = dict(slice=gen_slice, G1=gen_G1, G2=gen_G2_G3, G4=gen_G4_G5, G6=gen_G6, G7=gen_G7, G9=gen_G9, G10=gen_G10, G12=gen_G12, G13=gen_G13, G14=gen_G14, G15=gen_G15, G16=gen_G16, G17=gen_G17, G18=gen_G18, G19=gen_G19)

if __name__ == "__main__":
    rh.activate()
    which = sys.argv[1:] or list(ALL)
    for w in which:
        print("== %s" % w)
        ALL[w]()
