"""Where does a training step still take the atomic scatter (k_scatter_add_rows)?  Runs one eager step of BASELINE configs 1 / 3
(bench.other_config) with HipBackend.scatter_add_rows wrapped: prints every distinct Python call path that reaches it."""
import collections
import sys
import traceback

import torch

sys.path.insert(0, ".")
import bench
from temp_amd import backend as TB


def main():
    be = TB.get_backend()
    sites = collections.Counter()
    orig = type(be).scatter_add_rows

    def wrapped(self, src, idx, table):
        st = [f for f in traceback.extract_stack()[:-1] if "/temp_amd/" in f.filename]
        sites[" <- ".join("%s:%d %s" % (f.filename.split("/temp_amd/")[1], f.lineno, f.name) for f in reversed(st[-4:]))] += 1
        return orig(self, src, idx, table)

    type(be).scatter_add_rows = wrapped
    import argparse
    a = argparse.Namespace(no_graph=True)
    from temp_amd import _lib
    for name in sys.argv[1:] or ["config1_static", "config3_post_ensemble"]:
        sites.clear()
        r = bench.other_config(name, a, torch.device("cuda:0"), _lib.load(), 10)
        print(name, "launches/step", r["launches_per_step"])
        for k, v in sites.most_common():
            print("   %5d  %s" % (v, k))


if __name__ == "__main__":
    main()
