"""Helpers for the -m gpu tests: run the HIP path on a golden case / oracle problem."""
import warnings

import numpy as np

import fasta_python_amd as fa
from fasta_python_amd import stopping as fstop
from tests import helpers as H

TAGS = {
    "sparse_ls": lambda d: fa.Shrink(float(d["mu"])),
    "nnls": lambda d: fa.NonNeg(),
    "l1ball": lambda d: fa.L1Ball(float(d["mu"])),
    "linf": lambda d: fa.LinfProx(float(d["mu"])),
    "tv": lambda d: fa.TVDualBall(),
    "logistic": lambda d: fa.Shrink(float(d["mu"])),
}


def hip_operands(kind, data):
    """(A, loss, reg, x0) device-tagged operands for an oracle problem instance."""
    if kind == "tv":
        M, mu = data["M"], float(data["mu"])
        A = fa.GradDivMap(M.shape)
        loss = fa.LeastSquares(M / mu)
        x0 = np.zeros(M.shape + (2,))
    else:
        A = fa.DenseMatrixMap(np.asarray(data["A"]))
        loss = fa.LogisticLoss(data["b"]) if kind == "logistic" else fa.LeastSquares(data["b"])
        x0 = np.zeros(data["A"].shape[1])
    return A, loss, TAGS[kind](data), x0


def run_hip(kind, data, options, solver_seed, g_none=False):
    A, loss, reg, x0 = hip_operands(kind, data)
    try:
        o = H.resolve_options(options, fstop)
        o.pop("g_none", None)
        g, proxg = (None, None) if g_none else (reg.g, reg.prox)
        np.random.seed(solver_seed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return fa.fasta(A, A.H, loss.f, loss.gradf, g, proxg, x0, verbose=False, backend="hip", **o)
    finally:
        A.close()


def compare_histories(got, want_get, k, rtol, atol=0.0, fields=("residuals", "norm_residuals", "stepsizes", "objectives")):
    """Compare the first k entries of each history; returns the worst relative deviation seen."""
    worst = 0.0
    for field in fields:
        w = want_get(field)
        if w is None:
            continue
        g = getattr(got, field)
        assert g is not None, field
        hi = k + 1 if field in ("objectives", "function_hist") else k
        g, w = np.asarray(g)[:hi], np.asarray(w)[:hi]
        np.testing.assert_allclose(g, w, rtol=rtol, atol=atol, err_msg=field)
        denom = np.maximum(np.abs(w), 1e-300)
        worst = max(worst, float(np.max(np.abs(g - w) / denom)) if len(w) else 0.0)
    return worst
