"""Deterministic synthetic snapshot sets for measurement (SURVEY 8d / BASELINE.md section 3).

The reference's GDELT and ICEWS05-15 training files are not shipped (`.MISSING_LARGE_BLOBS`), so the
headline workload is a synthetic stand-in with GDELT's published shape (interpolation/gdelt/stat.txt:
500 entities, 20 relations, 366 timestamps; 2.74e6 quads / 366 = 7 475 edges per snapshot).

    topology seed 0 : node set per snapshot = uniform sample of N_ents without replacement;
                      dst ~ Zipf(1.0) and src ~ Zipf(1.0) independently over the node sample
                      (rank -> node by a per-snapshot permutation); rel ~ U[0, R); duplicates allowed;
                      no reverse edges.
    parameter seed 1: xavier-uniform with relu gain for embeddings / RGCN weights, U(+-1/sqrt(D)) for
                      the GRU (oracle.temp_oracle.init_model uses the same recipe).
"""
import numpy as np

from .snapshot import Snapshot

WORKLOADS = {
    # name:        N_ents, R,   E/snap, n/snap, T,  D,   B,   L,  bsz, module
    "S-gdelt":     (500, 20, 7475, 500, 366, 200, 100, 15, 8, "BiGRRGCN"),
    "S-icews14":   (7128, 230, 200, 227, 365, 200, 100, 8, 8, "GRRGCN"),
    "S-icews0515": (10488, 251, 92, 110, 64, 200, 100, 15, 8, "BiGRRGCN"),
    "S-hbm":       (1 << 20, 230, 1 << 24, 1 << 20, 32, 200, 100, 15, 1, "BiGRRGCN"),
    "S-tiny":      (64, 6, 300, 40, 24, 16, 8, 4, 3, "BiGRRGCN"),
    "S-gdelt-d128": (500, 20, 7475, 500, 366, 128, 128, 15, 8, "BiGRRGCN"),     # the reference's DEFAULT widths (utils/args.py:13-14,27: embed 128, 128 bases = 1 x 1 blocks)
}


def zipf_ranks(rng, n, size, s=1.0):
    """Samples in [0, n) with P(k) proportional to 1/(k+1)^s (inverse-CDF on the truncated law)."""
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, rng.random(size), side="left").astype(np.int64)


def make_snapshots(num_ents, num_rels, edges_per_snap, nodes_per_snap, num_times, seed=0):
    """{t: Snapshot} for t in 0..num_times-1."""
    rng = np.random.default_rng(seed)
    out = {}
    for t in range(num_times):
        if nodes_per_snap >= num_ents:
            ids = np.arange(num_ents, dtype=np.int64)
        else:
            ids = np.sort(rng.choice(num_ents, size=nodes_per_snap, replace=False)).astype(np.int64)
        n = ids.shape[0]
        perm_d, perm_s = rng.permutation(n), rng.permutation(n)
        dst = perm_d[zipf_ranks(rng, n, edges_per_snap)]
        src = perm_s[zipf_ranks(rng, n, edges_per_snap)]
        rel = rng.integers(0, num_rels, edges_per_snap)
        out[t] = Snapshot(n, src, dst, rel, ids)
    return out


def workload(name, seed=0):
    N, R, E, n, T, D, B, L, bsz, module = WORKLOADS[name]
    snaps = make_snapshots(N, R, E, n, T, seed)
    return dict(name=name, num_ents=N, num_rels=R, edges_per_snap=E, nodes_per_snap=n, num_times=T, D=D, B=B, L=L, bsz=bsz,
                module=module, snapshots=snaps)


def default_targets(num_times, L, bsz, rank=0, seed=3):
    """bsz target timestamps drawn uniformly without replacement (seeded, distinct per rank) from the
    timestamps whose forward AND backward windows are full -- what the reference's shuffled
    DataLoader over all timestamps produces (models/TKG_Module.py:162-179), so windows overlap as
    often as they do in real training and no more."""
    lo, hi = L - 1, num_times - L
    cand = np.arange(lo, max(hi, lo) + 1)
    rng = np.random.default_rng(seed + 1000 * rank)
    if len(cand) >= bsz:
        return [int(t) for t in rng.choice(cand, size=bsz, replace=False)]
    return [int(cand[i % len(cand)]) for i in range(bsz)]
