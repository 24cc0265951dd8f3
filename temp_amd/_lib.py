"""ctypes binding of libtemp_amd.so (the C ABI declared in include/temp_amd.h).

There is NO CPU fallback: if the shared library is missing or a symbol cannot be bound this module
raises, and every op in the package fails loudly.
"""
import ctypes
import threading
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtemp_amd.so")

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_vp = ctypes.c_void_p

ACT_NONE, ACT_RELU = 0, 1
GRU_TORCH, GRU_TYPE1 = 0, 1
CHUNK = 64
CHUNK_REL = 128


class TempEdgeView(ctypes.Structure):
    _fields_ = [("n_seg", ctypes.c_int32), ("n_edges", ctypes.c_int32), ("a", c_vp), ("b", c_vp),
                ("n_chunks", ctypes.c_int32), ("chunk_seg", c_vp), ("chunk_beg", c_vp), ("chunk_end", c_vp),
                ("chunk_slot", c_vp), ("n_partial", ctypes.c_int32), ("n_fix", ctypes.c_int32),
                ("fix_seg", c_vp), ("fix_slot", c_vp), ("fix_cnt", c_vp)]


class TempMembers(ctypes.Structure):
    _fields_ = [("n_members", ctypes.c_int32), ("max_nodes", ctypes.c_int32), ("max_edges", ctypes.c_int32),
                ("max_chunks", ctypes.c_int32 * 3), ("node_off", c_vp), ("edge_off", c_vp), ("chunk_off", c_vp), ("fix_off", c_vp)]


class TempGraph(ctypes.Structure):
    _fields_ = [("n_nodes", ctypes.c_int32), ("n_edges", ctypes.c_int32), ("nnorm", c_vp), ("in_deg", c_vp),
                ("out_deg", c_vp), ("by_dst", TempEdgeView), ("by_src", TempEdgeView), ("by_rel", TempEdgeView),
                ("members", TempMembers)]


ABI_VERSION = 2            # include/temp_amd.h: TEMP_ABI_VERSION (2: TempGraph.members, chain pipeline option)
OPT_MFMA_BF16X3, OPT_TN_SPLIT, OPT_RGCN_SCALAR, OPT_GEMM_STREAM, OPT_GRU_STREAM, OPT_RGCN_TILE, OPT_DEBUG, OPT_OVERLAP, OPT_GEMM_RESIDENT, OPT_MFMA_F16X2 = range(10)


class TempGruCellFwd(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("gi", c_vp), ("prev", c_vp), ("prev_idx", c_vp), ("dt", c_vp), ("w_hh", c_vp),
                ("b_hh", c_vp), ("h_out", c_vp), ("saved", c_vp)]


class TempGruCellBwd(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("saved", c_vp), ("dh_up", c_vp), ("d_prev_next", c_vp), ("next_idx", c_vp), ("dt", c_vp),
                ("w_hh", c_vp), ("dgi", c_vp), ("dgh", c_vp), ("decv", c_vp), ("d_prev", c_vp), ("no_prev", ctypes.c_int32)]


CHAIN_HAS_PREV = 1 << 30
CHAIN_TRACKS = 32
CHAIN_MAX_RNN = 4
CHAIN_MAX_UP = 8
CHAIN_MAX_STEPS = 64


class TempGruChain(ctypes.Structure):
    _fields_ = [("d", ctypes.c_int32), ("variant", ctypes.c_int32), ("n_panels", ctypes.c_int32), ("n_steps", ctypes.c_int32),
                ("max_steps", ctypes.c_int32), ("panel", c_vp), ("rows", c_vp), ("sinfo", c_vp), ("dt", c_vp), ("lambda_", ctypes.c_float),
                ("saved_plane", ctypes.c_size_t), ("n_rnn", ctypes.c_int32), ("packed", c_vp * CHAIN_MAX_RNN), ("b_hh", c_vp * CHAIN_MAX_RNN),
                ("gi_index", c_vp)]


class TempSubsampleJob(ctypes.Structure):
    _fields_ = [("n_nodes", ctypes.c_int32), ("n_edges", ctypes.c_int32), ("keep", ctypes.c_int32), ("seed", ctypes.c_uint64),
                ("parent", c_vp), ("child", c_vp), ("eid", c_vp),
                ("off_a", ctypes.c_int32 * 3), ("off_b", ctypes.c_int32 * 3), ("off_chunk_beg", ctypes.c_int32 * 3),
                ("off_chunk_end", ctypes.c_int32 * 3), ("off_chunk_seg", ctypes.c_int32 * 3), ("n_chunks", ctypes.c_int32 * 3),
                ("off_in_deg", ctypes.c_int32), ("off_out_deg", ctypes.c_int32), ("off_nnorm", ctypes.c_int32),
                ("keep_mask", c_vp), ("scratch", c_vp)]


class TempDropout(ctypes.Structure):
    _fields_ = [("p", ctypes.c_float), ("seed", ctypes.c_uint64)]


class TempAttn(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("D", ctypes.c_int32), ("heads", ctypes.c_int32), ("T", ctypes.c_int32),
                ("q", c_vp), ("ldq", ctypes.c_int32), ("kh", c_vp), ("vh", c_vp), ("ldh", ctypes.c_int32),
                ("kc", c_vp), ("vc", c_vp), ("ldc", ctypes.c_int32), ("idx", c_vp), ("decay", c_vp)]


class TempLinearProblem(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int32), ("A", c_vp), ("B", c_vp), ("C", c_vp)]


ASSEMBLE_PIECE = 4096
SCORE_KINDS = {"distmult": 0, "complex": 1}

# name -> (restype, argtypes); mirrors include/temp_amd.h one to one
_G = ctypes.POINTER(TempGraph)
_I, _F, _SZ = ctypes.c_int, ctypes.c_float, ctypes.c_size_t
SYMBOLS = {
    "temp_abi_version": (_I, []),
    "temp_error_string": (ctypes.c_char_p, [_I]),
    "temp_set_option": (_I, [_I, _I]),
    "temp_get_option": (_I, [_I]),
    "temp_scratch_refused": (ctypes.c_longlong, []),
    "temp_f16_launches": (ctypes.c_longlong, []),
    "temp_tile_launches": (ctypes.c_longlong, []),
    "temp_set_debug_buffer": (None, [c_vp, _SZ]),
    "temp_rgcn_fwd_workspace": (_SZ, [_G, _I]),
    "temp_rgcn_fwd": (_I, [_G, c_vp, c_vp, _I, _I, _I, _I, c_vp, c_vp, c_vp, _I, c_vp, c_vp, _SZ, c_vp, c_vp]),
    "temp_rgcn_bwd_workspace": (_SZ, [_G, _I, _I, _I, _I]),
    "temp_rgcn_bwd": (_I, [_G, c_vp, c_vp, c_vp, _I, _I, _I, _I, c_vp, c_vp, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, _SZ, c_vp, c_vp]),
    "temp_rgcn_bwd_dh": (_I, [_G, c_vp, c_vp, _I, _I, _I, _I, c_vp, c_vp, _I, c_vp, c_vp, c_vp, c_vp, _SZ, c_vp, c_vp]),
    "temp_rgcn_bwd_weights": (_I, [_G, c_vp, c_vp, c_vp, _I, _I, _I, _I, _I, c_vp, c_vp, c_vp, c_vp, _SZ, c_vp]),
    "temp_rgcn_table_fwd_workspace": (_SZ, [_G, _I, _I]),
    "temp_rgcn_table_fwd": (_I, [_G, c_vp, c_vp, _I, _I, _I, _I, _I, c_vp, c_vp, c_vp, _I, c_vp, c_vp, _SZ, c_vp, c_vp]),
    "temp_rgcn_table_bwd_workspace": (_SZ, [_G, _I, _I, _I, _I]),
    "temp_rgcn_table_bwd": (_I, [_G, c_vp, c_vp, c_vp, c_vp, _I, c_vp, c_vp, _I, _I, _I, _I, c_vp, c_vp, _I, _I, c_vp, c_vp, c_vp, c_vp,
                                 c_vp, _SZ, c_vp, c_vp]),
    "temp_rgcn_isolated_fwd": (_I, [_I, _I, c_vp, c_vp, c_vp, _I, c_vp, c_vp, c_vp]),
    "temp_rgcn_isolated_bwd_workspace": (_SZ, [_I, _I]),
    "temp_rgcn_isolated_bwd": (_I, [_I, _I, c_vp, c_vp, c_vp, c_vp, _I, _I, c_vp, c_vp, c_vp, c_vp, _SZ, c_vp, c_vp]),
    "temp_gru_fwd": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, _F, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_gru_bwd_workspace": (_SZ, [_I, _I, _I]),
    "temp_gru_bwd": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, _F, c_vp, c_vp, c_vp, c_vp, c_vp,
                          c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, _SZ, c_vp]),
    "temp_decay_rows": (_I, [_I, _I, c_vp, c_vp, _F, c_vp, c_vp]),
    "temp_gru_input_gates": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_gru_input_gates_multi": (_I, [_I, c_vp, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_gru_input_gates_gather_multi": (_I, [_I, c_vp, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_gru_cell_fwd": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, _F, c_vp, c_vp, c_vp, c_vp, _SZ, c_vp]),
    "temp_gru_cell_bwd": (_I, [_I, _I, _I, c_vp, _SZ, c_vp, c_vp, c_vp, c_vp, _F, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_gru_cell_fwd_multi": (_I, [_I, ctypes.POINTER(TempGruCellFwd), _I, _I, _F, _SZ, c_vp]),
    "temp_gru_cell_bwd_multi": (_I, [_I, ctypes.POINTER(TempGruCellBwd), _I, _I, _F, _SZ, c_vp]),
    "temp_gru_weight_grads_workspace": (_SZ, [_I, _I, _I]),
    "temp_gru_weight_grads": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, _SZ, c_vp]),
    "temp_gru_weight_grads_multi_workspace": (ctypes.c_size_t, [_I, c_vp, _I, _I]),
    "temp_gru_weight_grads_multi": (_I, [_I, c_vp, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "temp_gru_chain_supported": (_I, [_I]),
    "temp_gru_chain_pack_floats": (_SZ, [_I]),
    "temp_gru_chain_pack": (_I, [_I, c_vp, c_vp, c_vp]),
    "temp_gru_chain_pack_multi": (_I, [_I, _I, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), c_vp]),
    "temp_gru_chain_fwd": (_I, [ctypes.POINTER(TempGruChain), c_vp, c_vp, c_vp, c_vp]),
    "temp_gru_chain_bwd": (_I, [ctypes.POINTER(TempGruChain), c_vp, _I, ctypes.POINTER(c_vp), c_vp, c_vp, c_vp]),
    "temp_gru_chain_bwd_g4": (_I, [ctypes.POINTER(TempGruChain), c_vp, _I, ctypes.POINTER(c_vp), c_vp, c_vp]),
    "temp_gru_chain_keys_supported": (_I, [_I]),
    "temp_gru_chain_bwd_g4_keys": (_I, [ctypes.POINTER(TempGruChain), c_vp, _I, ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_vp]),
    "temp_gru_grads_g4_keys": (_I, [_I, c_vp, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "temp_gru_grads_g4_workspace": (ctypes.c_size_t, [_I, c_vp, _I]),
    "temp_gru_grads_g4": (_I, [_I, c_vp, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "temp_gather_rows": (_I, [_I, _I, c_vp, c_vp, c_vp, c_vp]),
    "temp_keys_cols_size": (_SZ, [_I]),
    "temp_absmax_keys": (_I, [_I, _I, c_vp, _I, c_vp, c_vp, c_vp]),
    "temp_gather_rows_keys": (_I, [_I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_linear_keys": (_I, [_I, _I, _I, c_vp, _I, c_vp, c_vp, _I, _I, c_vp, _I, c_vp]),
    "temp_gru_input_gates_gather_multi_keys": (_I, [_I, c_vp, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_scatter_add_rows": (_I, [_I, _I, c_vp, c_vp, c_vp, c_vp]),
    "temp_segment_sum_rows_workspace": (_SZ, [_I, _I, _I]),
    "temp_segment_sum_rows": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, _SZ, c_vp]),
    "temp_segment_sum_rows_relu": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, _SZ, c_vp]),
    "temp_linear": (_I, [_I, _I, _I, c_vp, _I, c_vp, _I, _I, c_vp, _I, c_vp]),
    "temp_linear_t": (_I, [_I, _I, _I, c_vp, _I, c_vp, _I, _I, c_vp, _I, c_vp]),
    "temp_linear_multi": (_I, [_I, ctypes.POINTER(TempLinearProblem), _I, _I, _I, _I, _I, _I, c_vp]),
    "temp_linear_tn_workspace": (_SZ, [_I, _I, _I]),
    "temp_linear_tn_multi_workspace": (_SZ, [_I, _I, _I, _I]),
    "temp_linear_tn_multi": (_I, [_I, ctypes.POINTER(TempLinearProblem), _I, _I, _I, _I, _I, c_vp, _SZ, c_vp]),
    "temp_linear_tn": (_I, [_I, _I, _I, c_vp, _I, c_vp, _I, c_vp, _I, c_vp, _SZ, c_vp]),
    "temp_gather_ce_fwd": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_gather_ce_bwd": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, _F, c_vp, c_vp, c_vp]),
    "temp_bilinear_query_fwd": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_bilinear_query_bwd": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_assemble_views": (_I, [_I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_subsample_views": (_I, [_I, ctypes.POINTER(TempSubsampleJob), c_vp]),
    "temp_corrupt_sample": (_I, [_I, _I, _I, ctypes.c_uint64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_filtered_rank": (_I, [_I, _I, _I, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "temp_sa_attn_fwd": (_I, [ctypes.POINTER(TempAttn), c_vp, c_vp, c_vp, c_vp]),
    "temp_sa_attn_bwd": (_I, [ctypes.POINTER(TempAttn), c_vp, c_vp, c_vp, c_vp, c_vp, _I, c_vp, c_vp, _I, c_vp, c_vp, _I, c_vp, _I, c_vp, c_vp, c_vp, c_vp]),
    "temp_copy_probe": (_I, [c_vp, c_vp, _SZ, c_vp]),
    "temp_trace_begin": (_I, [_I]),
    "temp_trace_end": (_I, [c_i32p, c_f32p, _I, c_i32p]),
    "temp_trace_kernel_name": (ctypes.c_char_p, [_I]),
}

_lib = None


class TempAmdError(RuntimeError):
    pass


def load():
    """Load the library once; raise TempAmdError when it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TempAmdError("libtemp_amd.so not found at %s -- run `python -m temp_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # PyDLL: the interpreter lock is NOT released around a call.  Every entry point only enqueues work (microseconds); with CDLL each of
    # the ~100 calls of a training step dropped the lock, a prefetch worker took it for a switch interval, and the step's issue code
    # crawled behind the planner threads (a convoy: 4.8-5.7 ms per fresh-batch step for 1.4 ms of issue code).  The host planner
    # library (_hostlib.py), whose calls run for hundreds of microseconds, keeps CDLL and runs without the lock.  TEMP_PYDLL=0: CDLL.
    lib = ctypes.CDLL(LIB_PATH) if os.environ.get("TEMP_PYDLL", "1") == "0" else ctypes.PyDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.temp_abi_version() != ABI_VERSION:
        raise TempAmdError("libtemp_amd.so ABI version %d != %d (stale .so or stale Python package)" % (lib.temp_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().temp_error_string(rc).decode()
        raise TempAmdError("%s failed: %s (code %d)" % (what, msg, rc))


# Resident, shared device objects (a snapshot's views, the true-set store) are created on first use by whichever thread gets
# there -- with several prefetch workers, each under its own HIP stream.  Creation is serialised by this lock and `publish()`
# drains the creating stream BEFORE the object becomes visible, so that any other stream may read it without an event.
class _CreateLock:
    """Re-entrant.  A prefetch worker never WAITS for it while holding the planning token (see _py_token below): the holder may be
    inside a planner call that takes the token back on return."""

    def __init__(self):
        self._l = threading.RLock()

    def __enter__(self):
        if not self._l.acquire(blocking=False):
            held = getattr(_coop, "held", False)
            if held:
                _py_token.release()
            self._l.acquire()
            if held:
                _py_token.acquire()
        return self

    def __exit__(self, *exc):
        self._l.release()
        return False


create_lock = _CreateLock()


def publish(device):
    import torch as _torch
    device = _torch.device(device)
    if device.type == "cuda":
        _torch.cuda.current_stream(device).synchronize()


# Cooperative hand-over of the interpreter lock between a training loop and its prefetch workers (prefetch.BatchPrefetcher).  The
# loop's launch code and a worker's planning code are both Python: run side by side they take the lock from each other every
# switch interval and each hand-over costs tens of microseconds (measured: 1.4 ms of launch code becomes 3.2 ms beside one worker).
# A worker therefore carries a gate (thread-local): `pause_point()` -- called between the stages of `prepare` -- parks the worker
# while the consumer is issuing a step, and the consumer opens the gate whenever it waits for a batch.  A thread without a gate
# (an inline `prepare`, a test) pays one attribute lookup.  Several workers would take the lock from EACH OTHER in the same way: a
# worker's planning Python runs under one token (a plain mutex: waiting for it sleeps, it does not spin on the interpreter lock),
# handed back around every call into the C++ planner library (_hostlib), so one worker's C++ runs beside another's Python.
_coop = threading.local()
_py_token = threading.Lock()


def coop_begin(gate):
    """A prefetch worker starts planning a batch: take the planning token; `gate` = the consumer's gate."""
    _coop.gate = gate
    _py_token.acquire()
    _coop.held = True


def coop_end():
    if getattr(_coop, "held", False):
        _coop.held = False
        _py_token.release()
    _coop.gate = None


def pause_point():
    g = getattr(_coop, "gate", None)
    if g is not None and not g.is_set():
        held = getattr(_coop, "held", False)
        if held:
            _py_token.release()
        g.wait()
        if held:
            _py_token.acquire()


def outside_token(fn):
    """fn(*args) with the planning token handed back for the duration of the call (the C++ planner calls: no Python runs inside)."""
    def call(*args):
        if getattr(_coop, "held", False):
            _py_token.release()
            try:
                return fn(*args)
            finally:
                _py_token.acquire()
        return fn(*args)
    return call


_STAGE_BYTES = 32 << 20
_stage = threading.local()


def to_device(host, device):
    """Host tensor / numpy array -> `device`.  On a GPU the bytes are staged in a per-thread pinned ring buffer and the
    copy is stream-ordered (non_blocking): no host-side wait per upload.  A pageable copy blocks the calling thread until
    the stream reaches it -- tens of microseconds each even when idle, and a window batch has dozens of small index vectors.
    The ring has two halves: when one is full, an event is recorded on every stream that carried copies out of it, and the
    half is written again only after those events (a full cycle later: they have long passed).  A device-wide synchronize at
    that point made the host wait for the training step in flight once every few batches."""
    import numpy as _np
    import torch as _torch
    t = _torch.from_numpy(_np.ascontiguousarray(host)) if isinstance(host, _np.ndarray) else host
    device = _torch.device(device)
    nbytes = t.numel() * t.element_size()
    if device.type != "cuda" or nbytes == 0 or t.is_cuda or nbytes > _STAGE_BYTES // 4 or not t.is_contiguous():
        return t.to(device)
    st = _stage
    buf = getattr(st, "buf", None)
    half_bytes = _STAGE_BYTES // 2
    if buf is None:
        buf = st.buf = _torch.empty(_STAGE_BYTES, dtype=_torch.uint8, pin_memory=True)
        st.np = buf.numpy()
        st.off, st.half = 0, 0
        st.streams = {}                      # streams that carried copies out of the current half
        st.events = [[], []]                 # per half: events behind its last copies
    if st.off + nbytes > (st.half + 1) * half_bytes:
        evs = []
        for s in st.streams.values():
            ev = _torch.cuda.Event()
            ev.record(s)
            evs.append(ev)
        st.events[st.half] = evs
        st.streams = {}
        st.half ^= 1
        st.off = st.half * half_bytes
        for ev in st.events[st.half]:
            ev.synchronize()
        st.events[st.half] = []
    off = st.off
    st.off += (nbytes + 255) & ~255
    st.np[off:off + nbytes] = t.numpy().reshape(-1).view(_np.uint8)      # plain single-threaded memcpy
    out = _torch.empty(t.shape, dtype=t.dtype, device=device)
    out.copy_(buf[off:off + nbytes].view(t.dtype).view(t.shape), non_blocking=True)
    cur = _torch.cuda.current_stream(device)
    st.streams[cur.cuda_stream] = cur
    return out
