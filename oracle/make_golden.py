"""Capture golden vectors from the REAL reference core.  Runs only in the build container.

    PYTHONPATH=/root/reference MPLBACKEND=Agg python -m oracle.make_golden

imports `fasta.fasta`, `fasta.linalg.LinearMap`, `fasta.proximal`, `fasta.stopping` from
/root/reference (the core imports; the example modules do not -- SURVEY.md section 0.1), builds
problem instances with the restated recipes of oracle/problems.py but using the REFERENCE's prox
functions / stop rules / LinearMap inside the closures, runs the reference solver, and writes
inputs + every Convergence field to tests/golden/<case>.npz plus KAT files.

The fixtures are data (inputs and expected outputs).  The reference itself never travels.
"""

import json
import os
import sys

import numpy as np
from numpy import linalg as la

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")

MODES = {
    "adaptive": dict(adaptive=True, accelerate=False),
    "accelerated": dict(adaptive=False, accelerate=True),
    "plain": dict(adaptive=False, accelerate=False),
}
TEST_MODES_OPTS = dict(tolerance=1e-5, evaluate_objective=True)     # examples/__init__.py:63-91


def case_table():
    """(name, kind, construct-kwargs, problem-seed, solver-seed, fasta options)."""
    cases = []
    for mode, mo in MODES.items():
        o = dict(TEST_MODES_OPTS, **mo)
        cases.append((f"sparse_ls_64x128_{mode}", "sparse_ls", dict(M=64, N=128, K=5), 11, 101, o))
        cases.append((f"nnls_128x64_{mode}", "nnls", dict(M=128, N=64, K=5), 12, 102, o))
        cases.append((f"l1ball_64x128_{mode}", "l1ball", dict(M=64, N=128, K=5), 13, 103, o))
        cases.append((f"linf_96x96_{mode}", "linf", dict(M=96, N=96, mu=0.05), 14, 104, o))
        cases.append((f"tv_32x32_{mode}", "tv", dict(H=32, W=32, square=8), 15, 105,
                      dict(o, max_iters=300)))
        cases.append((f"logistic_100x160_{mode}", "logistic", dict(M=100, N=160, K=4, mu=4.0), 19, 106,
                      dict(o, max_iters=400)))
    # option coverage on the sparse-LS instance
    base = dict(tolerance=1e-5)
    cases += [
        ("sparse_ls_opt_nobacktrack", "sparse_ls", dict(M=64, N=128, K=5), 11, 111,
         dict(base, backtrack=False)),
        ("sparse_ls_opt_given_L_tau", "sparse_ls", dict(M=64, N=128, K=5), 11, 112,
         dict(base, L=1.0, tau0=0.15)),
        ("sparse_ls_opt_lone_tau0_is_overwritten", "sparse_ls", dict(M=64, N=128, K=5), 11, 113,
         dict(base, tau0=0.15)),
        ("sparse_ls_opt_record_func", "sparse_ls", dict(M=64, N=128, K=5), 11, 114,
         dict(base, record_iterates=True, func="l1norm", evaluate_objective=True)),
        ("sparse_ls_opt_accel_norestart", "sparse_ls", dict(M=64, N=128, K=5), 11, 115,
         dict(base, adaptive=False, accelerate=True, restart=False, max_iters=200)),
        ("sparse_ls_opt_accel_adaptive", "sparse_ls", dict(M=64, N=128, K=5), 11, 116,
         dict(base, adaptive=True, accelerate=True)),
        ("sparse_ls_opt_window3_shrink", "sparse_ls", dict(M=64, N=128, K=5, normalise=False), 11, 117,
         dict(base, window=3, stepsize_shrink=0.5, max_backtracks=4, max_iters=150)),
        ("sparse_ls_unnormalised_backtracks", "sparse_ls", dict(M=64, N=128, K=5, normalise=False), 16, 118,
         dict(base, L=1.0, tau0=1.0, max_iters=200)),
        ("nnls_under_first40", "nnls", dict(M=64, N=128, K=5), 17, 119,
         dict(tolerance=0.0, max_iters=40)),
        ("sparse_ls_gradient_descent_g_none", "sparse_ls", dict(M=128, N=64, K=5), 18, 120,
         dict(base, g_none=True, max_iters=300)),
    ]
    for rule in ("residual", "norm_residual", "ratio_residual", "hybrid_residual"):
        cases.append((f"sparse_ls_stop_{rule}", "sparse_ls", dict(M=64, N=128, K=5), 11, 121,
                      dict(tolerance=1e-4, stop_rule=rule)))
    # BASELINE config 1 (512x1024, reference's literal spectral normalisation): inputs by seed only
    for mode, mo in MODES.items():
        cases.append((f"c1_sparse_ls_512x1024_{mode}", "sparse_ls", dict(M=512, N=1024, K=10), 21, 201,
                      dict(TEST_MODES_OPTS, **mo)))
    return cases


def build_with_reference_ops(kind, ckw, ref):
    """Problem instance whose closures call the REFERENCE's prox functions."""
    from oracle import problems as pr
    P = getattr(pr, {"sparse_ls": "sparse_least_squares", "nnls": "nn_least_squares",
                     "l1ball": "l1_ball_lasso", "linf": "linf_regularised",
                     "tv": "tv_denoising", "logistic": "sparse_logistic"}[kind])(**ckw)
    d = P.data
    if kind in ("sparse_ls", "logistic"):
        P.proxg = lambda x, t: ref.proximal.shrink(x, t * d["mu"])
    elif kind == "l1ball":
        P.proxg = lambda x, t: ref.proximal.project_L1_ball(x, d["mu"])
    elif kind == "linf":
        P.proxg = lambda x, t: ref.proximal.project_Linf_ball(x, t * d["mu"])
    return P


def main():
    sys.path.insert(0, "/root/reference")
    os.environ.setdefault("MPLBACKEND", "Agg")
    import fasta as ref                                   # the reference core
    assert ref.__file__.startswith("/root/reference"), ref.__file__
    sys.path.insert(0, ROOT)
    os.makedirs(GOLDEN, exist_ok=True)

    index = {}
    for name, kind, ckw, pseed, sseed, opts in case_table():
        np.random.seed(pseed)
        P = build_with_reference_ops(kind, ckw, ref)
        o = dict(opts)
        if isinstance(o.get("stop_rule"), str):
            o["stop_rule"] = getattr(ref.stopping, o["stop_rule"])
        if o.get("func") == "l1norm":
            o["func"] = lambda x: la.norm(x.ravel(), 1)
        g, proxg = P.g, P.proxg
        if o.pop("g_none", False):
            g, proxg = None, None
        if kind == "tv":
            A = ref.linalg.LinearMap(P.A, P.At, P.x0.shape, P.x0.shape[:-1])
        else:
            A = ref.linalg.LinearMap.from_matrix(P.data["A"])
        np.random.seed(sseed)                             # fasta() draws 2 randn from the global RNG
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = ref.fasta(A, P.f, P.gradf, g, proxg, P.x0, verbose=False, **o)
        out = dict(residuals=c.residuals, norm_residuals=c.norm_residuals, stepsizes=c.stepsizes,
                   backtracks=np.int64(c.backtracks), iteration_count=np.int64(c.iteration_count),
                   solution=c.solution)
        for opt in ("objectives", "iterates", "function_hist"):
            if getattr(c, opt) is not None:
                out[opt] = getattr(c, opt)
        big = name.startswith("c1_")
        inputs = {} if big else {f"in_{k}": np.asarray(v) for k, v in P.data.items() if v is not None}
        meta = dict(name=name, kind=kind, construct=ckw, problem_seed=pseed, solver_seed=sseed,
                    options=opts, inputs_by_seed=big,
                    numpy=np.__version__)
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), meta=json.dumps(meta), **inputs, **out)
        index[name] = dict(iters=int(c.iteration_count), backtracks=int(c.backtracks))
        print(f"{name:48s} iters={int(c.iteration_count):4d} backtracks={int(c.backtracks):3d}")

    # known-answer tests for prox operators and stop rules
    x = np.array([3, -1, .5, -4, 0, 2.0])
    rng = np.random.RandomState(5)
    xr = rng.randn(257)
    kat = {
        "x": x, "xr": xr,
        "shrink_t1": ref.proximal.shrink(x, 1.0),
        "linf_t1": ref.proximal.project_Linf_ball(x, 1.0),
        "linf_t4": ref.proximal.project_Linf_ball(x, 4.0),
        "linf_t10p5": ref.proximal.project_Linf_ball(x, 10.5),
        "linf_t11": ref.proximal.project_Linf_ball(x, 11.0),
        "l1_t4": ref.proximal.project_L1_ball(x, 4.0),
        "l1_t1": ref.proximal.project_L1_ball(x, 1.0),
        "l1_t10p5": ref.proximal.project_L1_ball(x, 10.5),
        "shrink_r": ref.proximal.shrink(xr, 0.3),
        "linf_r": ref.proximal.project_Linf_ball(xr, 7.0),
        "l1_r": ref.proximal.project_L1_ball(xr, 7.0),
    }
    np.savez_compressed(os.path.join(GOLDEN, "kat_prox.npz"), **kat)
    stop = {}
    for rule in ("residual", "norm_residual", "ratio_residual", "hybrid_residual"):
        fn = getattr(ref.stopping, rule)
        for args in ((0, 1e-6, 1.0, 1.0, 1e-5), (0, 1.0, 1e-6, 1.0, 1e-5), (0, 1.0, 1.0, 1.0, 1e-5),
                     (3, 2e-6, 0.5, 1.0, 1e-5), (3, 0.5, 0.5, 1e9, 1e-5)):
            stop[f"{rule}|{json.dumps(args)}"] = bool(fn(*args))
    with open(os.path.join(GOLDEN, "kat_stopping.json"), "w") as fh:
        json.dump(stop, fh, indent=1, sort_keys=True)
    with open(os.path.join(GOLDEN, "index.json"), "w") as fh:
        json.dump(index, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
