"""Host-logic tests of the drop-in encoder classes (temp_amd.{rgcn,rrgcn,birrgcn}) on CPU.

The HIP kernels cannot run here, so a TEST-ONLY backend (tests/cpu_backend.py) that follows the
C-ABI contract through the same chunked edge views is installed; what is verified is everything
around the kernels: graph views, autograd wiring, the reference's aliasing / return conventions,
state_dict keys.  Expected values are the golden vectors recorded from the reference itself.
The same assertions run against the real kernels in tests/test_gpu_parity.py (-m gpu)."""
import argparse

import numpy as np
import pytest
import torch

import temp_amd
from temp_amd import backend as TB
from temp_amd.snapshot import Snapshot
from tests.cpu_backend import CpuTestBackend
from tests.encoder_cases import (check_G2, check_G4, check_G6, check_G7)


@pytest.fixture(autouse=True)
def cpu_backend():
    TB.set_backend(CpuTestBackend())
    yield
    TB.set_backend(None)


def test_product_has_no_cpu_path():
    TB.set_backend(None)
    h = torch.zeros(4, 8)
    with pytest.raises(Exception):
        be = TB.HipBackend()            # loads the library (fine) ...
        be.rgcn_isolated_fwd(h, torch.zeros(8, 8), None, 0)   # ... but CPU tensors are refused


def test_G2_layer():
    check_G2(torch.device("cpu"))


def test_G4_grrgcn_layer():
    check_G4(torch.device("cpu"))


def test_G6_rrgcn():
    check_G6(torch.device("cpu"))


def test_G7_birrgcn():
    check_G7(torch.device("cpu"))
