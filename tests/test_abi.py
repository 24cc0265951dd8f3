"""The C-ABI shared library loads on a machine without a GPU and exports every symbol that
include/temp_amd.h declares (no compute calls here)."""
import os
import re

from temp_amd import _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "temp_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(temp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libtemp_amd.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names, "temp_amd/_lib.py binds a different set than the header declares"


def test_abi_version_and_error_strings():
    lib = _lib.load()
    assert lib.temp_abi_version() == 2
    assert lib.temp_error_string(0) == b"ok"
    assert b"workspace" in lib.temp_error_string(3)
    assert lib.temp_trace_kernel_name(0).startswith(b"k_rgcn_agg")
