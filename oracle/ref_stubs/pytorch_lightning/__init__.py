"""Inert stand-in for `pytorch_lightning==0.5.2` (harness only, no arithmetic).

TEST INFRASTRUCTURE ONLY -- lets `oracle/gen_golden.py` import the reference's
TKG_Module subclasses in the build container.  See oracle/README.md.
"""
from . import callbacks, logging, root_module  # noqa: F401


def data_loader(fn):
    return fn


class Trainer:
    def __init__(self, *a, **k):
        pass
