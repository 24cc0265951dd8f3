"""temp_amd -- MI355X-native TeMP snapshot encoder (RGCN message passing + GRU/BiGRU window).

The compute path is hand-written HIP for gfx950 behind the C ABI in include/temp_amd.h
(libtemp_amd.so, built in-tree by `python -m temp_amd.build`); this package is the host-side
mirror of the reference's encoder interface.  No CPU fallback exists.
"""
from . import _lib  # noqa: F401
from .snapshot import Snapshot, batch  # noqa: F401
from .rgcn import RGCN, RGCNLayer  # noqa: F401
from .rrgcn import GRRGCNLayer, RRGCN, RRGCNLayer  # noqa: F401
from .birrgcn import BiGRRGCNLayer, BiRRGCN, BiRRGCNLayer  # noqa: F401
from .gru_cell import GRUCell  # noqa: F401

__version__ = "0.1.0"
