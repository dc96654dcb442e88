"""The NumPy oracle must reproduce the reference core BIT-FOR-BIT on every captured fixture
(histories, counters and the returned best iterate) -- this is what pins parity (prompt item 3,
SURVEY.md section 8(c))."""
import json
import os

import numpy as np
import pytest

from oracle import fasta_np as fo
from tests import helpers as H


@pytest.mark.parametrize("name", H.golden_cases())
def test_oracle_matches_reference_bitwise(name):
    meta, z = H.load_case(name)
    c = H.run_oracle(meta, z)
    assert c.iteration_count == int(z["iteration_count"])
    assert c.backtracks == int(z["backtracks"])
    for field in H.HISTORY_FIELDS:
        if field in z.files:
            got = getattr(c, field)
            assert got is not None, field
            assert np.array_equal(got, z[field], equal_nan=True), field
        else:
            assert getattr(c, field) is None, field
    assert np.array_equal(c.solution, z["solution"])


def test_pass_counts_match_survey_probe():
    # A passes = iters + backtracks + 3, AH passes = iters + 3
    meta, z = H.load_case("tv_32x32_adaptive")
    c = H.run_oracle(meta, z)
    assert c.passes["A"] == c.iteration_count + c.backtracks + 3
    assert c.passes["AH"] == c.iteration_count + 3


def test_prox_known_answers(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat_prox.npz"))
    x, xr = k["x"], k["xr"]
    assert np.array_equal(fo.shrink(x, 1.0), k["shrink_t1"])
    assert np.array_equal(fo.shrink(x, 1.0), [2, -0.0, 0, -3, 0, 1])
    for t, key in ((1.0, "linf_t1"), (4.0, "linf_t4"), (10.5, "linf_t10p5"), (11.0, "linf_t11")):
        assert np.array_equal(fo.prox_linf(x, t), k[key]), key
    assert np.array_equal(fo.prox_linf(x, 1.0), [3, -1, .5, -3, 0, 2])         # P2
    assert np.allclose(fo.prox_linf(x, 4.0), [5 / 3, -1, .5, -5 / 3, 0, 5 / 3], rtol=0, atol=1e-15)
    assert not fo.prox_linf(x, 11.0).any()
    for t, key in ((4.0, "l1_t4"), (1.0, "l1_t1"), (10.5, "l1_t10p5")):
        assert np.array_equal(fo.project_l1(x, t), k[key]), key
    assert abs(np.abs(fo.project_l1(x, 4.0)).sum() - 4.0) < 1e-14              # P3
    assert np.array_equal(fo.shrink(xr, 0.3), k["shrink_r"])
    assert np.array_equal(fo.prox_linf(xr, 7.0), k["linf_r"])
    assert np.array_equal(fo.project_l1(xr, 7.0), k["l1_r"])


def test_stop_rule_known_answers(golden_dir):
    with open(os.path.join(golden_dir, "kat_stopping.json")) as fh:
        table = json.load(fh)
    for key, want in table.items():
        rule, args = key.split("|")
        assert bool(getattr(fo, rule)(*json.loads(args))) == want, key
    # SURVEY 8(a) S1-4 literal values
    assert fo.residual(0, 1e-6, 1, 1, 1e-5) and fo.norm_residual(0, 1, 1e-6, 1, 1e-5)
    assert fo.ratio_residual(0, 1e-6, 1, 1, 1e-5) and not fo.hybrid_residual(0, 1, 1, 1, 1e-5)


def test_linear_map_contract():
    rng = np.random.RandomState(0)
    M = rng.randn(5, 7)
    A = fo.LinearMap.from_matrix(M)
    x, y = rng.randn(7), rng.randn(5)
    assert np.array_equal(A(x), M @ x) and np.array_equal(A.H(y), M.T @ y)
    assert A.Vshape == (7,) and A.Wshape == (5,) and A.H.Vshape == (5,)
    with pytest.raises(AssertionError):
        A(y)                                             # linalg.py:58 shape assert
    with pytest.raises(AssertionError):
        fo.LinearMap.from_matrix(x)                      # linalg.py:40 ndim assert
    I = fo.LinearMap.identity((3, 2))
    v = rng.randn(3, 2)
    assert I(v) is v


def test_tv_operators_are_adjoint():
    from oracle import problems as pr
    rng = np.random.RandomState(1)
    X, Y = rng.randn(9, 11), rng.randn(9, 11, 2)
    assert abs(np.vdot(pr.div(Y), X) - np.vdot(Y, pr.grad(X))) < 1e-12
