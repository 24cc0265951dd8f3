"""Helpers shared by the golden-fixture tests (fixtures were produced by oracle/gen_golden.py from
the reference's own modules; parameters are regenerated from the stored seed and verified
against the stored checksum)."""
import os

import numpy as np
import torch

from oracle import temp_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if dtype is not None else t


def graph_from(z, prefix=""):
    return O.SnapGraph(int(z[prefix + "n"]), z[prefix + "src"], z[prefix + "dst"], z[prefix + "rel"], z[prefix + "ids"],
                       z[prefix + "nnorm"], z[prefix + "enorm"] if (prefix + "enorm") in z.files else None)


def checksum(model):
    return float(sum(v.double().abs().sum().item() for v in O.leaf_tensors(model).values()))


def layer_params(rng, D, B, R2, Tn, bias):
    """Mirror of gen_golden._layer_params (same draw order)."""
    s = D // B
    return dict(weight=O._xavier(rng, R2, B * s * s), loop_weight=O._xavier(rng, D, D), time_embed=O._xavier(rng, Tn, D),
                h_bias=(torch.from_numpy(rng.uniform(-0.5, 0.5, D).astype(np.float32)) if bias else None))


def slice_graphs():
    """Oracle-side rebuild of the per-timestamp train/valid/test snapshots from the committed
    ICEWS14 slice, following utils/dataset.py:151-232 (node set = union over the three splits,
    `np.unique` order, train graph without reverse edges, norm = 1/in_deg)."""
    z = load("icews14_slice")
    times = [int(t) for t in z["times"]]
    out = {"train": {}, "valid": {}, "test": {}}
    quads = {k: z[k] for k in ("train", "valid", "test")}
    for t in times:
        trip = {k: q[q[:, 3] == t][:, :3].astype(np.int64) for k, q in quads.items()}
        total = np.concatenate([trip["train"], trip["valid"], trip["test"]], axis=0)
        uniq, edges = np.unique((total[:, 0], total[:, 2]), return_inverse=True)
        src, dst = np.reshape(edges, (2, -1))
        a, b = len(trip["train"]), len(trip["train"]) + len(trip["valid"])
        for k, sl in (("train", slice(0, a)), ("valid", slice(a, b)), ("test", slice(b, None))):
            out[k][t] = O.SnapGraph(len(uniq), src[sl], dst[sl], total[sl, 1], uniq)
    return int(z["num_ents"]), int(z["num_rels"]), times, out


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    if a.numel() == 0:
        return
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError("%s: %d/%d elements out of tolerance (rtol=%g atol=%g); worst |err|=%.3e at flat %d "
                             "(got %.8g, want %.8g); max|want|=%.3e" %
                             (what, int(bad.sum()), a.numel(), rtol, atol, float(err.view(-1)[i]), i,
                              float(a.view(-1)[i]), float(b.view(-1)[i]), float(b.abs().max())))
