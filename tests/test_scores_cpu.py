"""temp_amd.scores (the reference's preserved scorer signatures) against G9_scores.npz, recorded from the reference's own
utils/scores.py -- the product module itself, not the oracle's copy (that one is pinned in tests/test_oracle_golden.py)."""
import pytest
import torch

from temp_amd import scores as SC
from tests.golden_util import T, assert_close, load


@pytest.mark.parametrize("name", ["distmult", "complex", "transE"])
def test_scores_module_vs_reference_golden(name):
    z = load("G9_scores")
    s, r, o, cand = (T(z[k]) for k in ("s", "r", "o", "cand"))
    fn = getattr(SC, name)
    assert_close(fn(s, r, o), z[name + "_single"], 1e-5, 1e-6, name + " single")
    assert_close(fn(s, r, cand, mode="tail"), z[name + "_tail"], 1e-5, 1e-6, name + " tail")
    assert_close(fn(cand, r, o, mode="head"), z[name + "_head"], 1e-5, 1e-6, name + " head")


@pytest.mark.parametrize("name", ["distmult", "complex"])
def test_folded_query_reproduces_reference_scores(name):
    """score(candidate) = <bilinear_query(known, r), candidate> -- the formulation the fused loss and evaluate() use."""
    z = load("G9_scores")
    s, r, o, cand = (T(z[k]) for k in ("s", "r", "o", "cand"))
    q_tail = SC.bilinear_query(name, s, r, "tail")
    q_head = SC.bilinear_query(name, o, r, "head")
    assert_close((q_tail.unsqueeze(1) * cand).sum(-1), z[name + "_tail"], 1e-5, 1e-6, name + " tail via query")
    assert_close((q_head.unsqueeze(1) * cand).sum(-1), z[name + "_head"], 1e-5, 1e-6, name + " head via query")
    assert SC.bilinear_query("transE", s, r, "tail") is None
