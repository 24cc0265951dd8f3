"""Quadruple files -> per-timestamp snapshots (the input contract of the hot path).

Follows the reference's interpolation builder (utils/dataset.py:12-48,151-232,235-251): for every
timestamp the node set is the union of the entities of its train, valid and test triples
(`np.unique` order = ascending global id), local ids index that set, and the train / valid / test
graphs share it; no reverse edges are added; norm = 1/in_degree (inf -> 0).
"""
import os

import numpy as np

from .snapshot import Snapshot


def load_quadruples(dataset_path, *file_names):
    """(quads (N,4) int64 [head, rel, tail, time], sorted unique times)."""
    rows = []
    for fn in file_names:
        with open(os.path.join(dataset_path, fn), 'r') as fr:
            for line in fr:
                sp = line.split()
                rows.append((int(sp[0]), int(sp[1]), int(sp[2]), int(sp[3])))
    quads = np.asarray(rows, dtype=np.int64).reshape(-1, 4)
    return quads, np.unique(quads[:, 3])


def get_total_number(dataset_path, file_name="stat.txt"):
    with open(os.path.join(dataset_path, file_name), 'r') as fr:
        sp = fr.readline().split()
        return int(sp[0]), int(sp[1])


def build_interpolation_snapshots(train, valid, test, times=None):
    """-> (graph_dict_train, graph_dict_val, graph_dict_test): {time: Snapshot}."""
    splits = [np.asarray(q, dtype=np.int64).reshape(-1, 4) for q in (train, valid, test)]
    if times is None:
        times = np.unique(np.concatenate([q[:, 3] for q in splits]))
    dicts = ({}, {}, {})
    for t in times:
        t = int(t)
        trip = [q[q[:, 3] == t][:, :3] for q in splits]
        total = np.concatenate(trip, axis=0)
        uniq, inv = np.unique((total[:, 0], total[:, 2]), return_inverse=True)
        src, dst = np.reshape(inv, (2, -1))
        a = len(trip[0])
        b = a + len(trip[1])
        for d, sl in zip(dicts, (slice(0, a), slice(a, b), slice(b, None))):
            d[t] = Snapshot(len(uniq), src[sl], dst[sl], total[sl, 1], uniq)
    return dicts


def build_interpolation_graphs(args):
    """Entry point with the reference's name/signature (utils/dataset.py:268-305), reading
    `args.dataset`/{train,valid,test}.txt; nothing is pickled next to the dataset."""
    tr, _ = load_quadruples(args.dataset, 'train.txt')
    va, _ = load_quadruples(args.dataset, 'valid.txt')
    te, _ = load_quadruples(args.dataset, 'test.txt')
    return build_interpolation_snapshots(tr, va, te)
