"""CPU tier for the product's GENERIC host loop (fasta_python_amd/generic.py): operands that cannot run inside a
kernel -- the reference's own lambda forms (examples/sparse_least_squares.py:41-44 and siblings), a callable pair,
`A=None`, a host LinearMap -- go through `import fasta; fasta.fasta(...)` and must reproduce every fixture captured
from the reference BIT FOR BIT.  Nothing here imports oracle/: the closures are built from the product's own
`fasta.proximal` / `fasta.linalg` / `fasta.stopping`, exactly as a user of the reference would write them."""
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
from numpy import linalg as la

import fasta
from fasta import proximal, stopping
from fasta.linalg import LinearMap, LinearOperator
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_style_operands(kind, d, call_form):
    """(A, At, f, gradf, g, proxg, x0) written the way the reference's example modules write them."""
    if kind == "tv":
        from fasta.examples.tv_denoising import div, grad
        M, mu = d["M"], float(d["mu"])
        f = lambda Z: .5 * la.norm((Z - M / mu).ravel()) ** 2                 # tv_denoising.py:85-87
        gradf = lambda Z: Z - M / mu
        g = lambda Y: 0

        def proxg(Y, t):                                                       # tv_denoising.py:89-96
            norms = la.norm(Y, axis=Y.ndim - 1)
            norms = np.maximum(norms, 1)
            return Y / norms[..., np.newaxis]
        Y0 = np.zeros(M.shape + (2,))
        if call_form == "linear_map":
            return LinearMap(div, grad, Y0.shape, M.shape), None, f, gradf, g, proxg, Y0
        return div, grad, f, gradf, g, proxg, Y0                               # tv_denoising.py:99: bare functions
    A, b = d["A"], d["b"]
    if kind == "logistic":
        f = lambda z: np.sum(np.log(1 + np.exp(z)) - (b == 1) * z)             # sparse_logistic.py:47-48
        gradf = lambda z: -b / (1 + np.exp(b * z))
    else:
        f = lambda z: .5 * la.norm((z - b).ravel()) ** 2                       # sparse_least_squares.py:41-42
        gradf = lambda z: z - b
    mu = float(d["mu"]) if "mu" in d else None
    if kind in ("sparse_ls", "logistic"):
        g = lambda x: mu * la.norm(x.ravel(), 1)                               # sparse_least_squares.py:43-44
        proxg = lambda x, t: proximal.shrink(x, t * mu)
    elif kind == "nnls":
        g = lambda x: 0                                                        # nn_least_squares.py:41-42
        proxg = lambda x, t: np.maximum(x, 0)
    elif kind == "l1ball":
        g = lambda x: 0                                                        # lasso.py:44-45
        proxg = lambda x, t: proximal.project_L1_ball(x, mu)
    elif kind == "linf":
        g = lambda x: mu * la.norm(x, np.inf)                                  # democratic_representation.py:41-42
        proxg = lambda x, t: proximal.project_Linf_ball(x, t * mu)
    x0 = np.zeros(A.shape[1])
    if call_form == "linear_map":                                              # 6-argument core form, fasta/__init__.py:38-40
        return LinearMap(lambda x: A @ x, lambda y: A.T @ y, (A.shape[1],), (A.shape[0],)), None, f, gradf, g, proxg, x0
    if call_form == "callables":
        return (lambda x: A @ x), (lambda y: A.T @ y), f, gradf, g, proxg, x0
    return A, A.T, f, gradf, g, proxg, x0                                      # sparse_least_squares.py:46: matrices


def run_generic(name, call_form):
    meta, z = H.load_case(name)
    d = H.case_data(meta, z)
    A, At, f, gradf, g, proxg, x0 = reference_style_operands(meta["kind"], d, call_form)
    o = H.resolve_options(meta["options"], stopping)
    if o.pop("g_none", False):
        g, proxg = None, None
    np.random.seed(meta["solver_seed"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if At is None:
            c = fasta.fasta(A, f, gradf, g, proxg, x0, verbose=False, **o)
        else:
            c = fasta.fasta(A, At, f, gradf, g, proxg, x0, verbose=False, **o)
    return z, c


def assert_bitwise(z, c):
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


@pytest.mark.parametrize("name", H.golden_cases())
def test_reference_lambda_forms_match_the_reference_bitwise(name):
    """7-argument example form: raw matrices (or the bare div/grad functions) + lambdas."""
    assert_bitwise(*run_generic(name, "as_examples"))


@pytest.mark.parametrize("call_form", ["linear_map", "callables"])
@pytest.mark.parametrize("name", ["sparse_ls_64x128_adaptive", "nnls_128x64_accelerated", "tv_32x32_adaptive", "l1ball_64x128_plain",
                                  "sparse_ls_gradient_descent_g_none"])
def test_other_operator_forms_match_the_reference_bitwise(name, call_form):
    """6-argument core form with a host LinearMap; 7-argument form with a callable pair."""
    assert_bitwise(*run_generic(name, call_form))


def test_identity_operator_from_none():
    """`fasta(None, None, ...)` (svm.py:74, nn_factorization.py:63): A is the identity on x0's space.
    Box-constrained quadratic, the svm.py:66-71 recipe: the loop must equal the same call with an explicit identity map."""
    rng = np.random.RandomState(3)
    D, l, C = rng.randn(40, 12), np.sign(rng.randn(40)), 0.7
    f = lambda y: .5 * la.norm((D.T @ (l * y)).ravel()) ** 2 - np.sum(y)
    gradf = lambda y: l * (D @ (D.T @ (l * y))) - 1
    g = lambda y: 0
    proxg = lambda y, t: np.minimum(np.maximum(y, 0), C)
    y0 = np.zeros(40)
    out = []
    for A, At in ((None, None), (LinearMap.identity((40,)), None)):
        np.random.seed(4)
        args = (A, At, f, gradf, g, proxg, y0) if At is None and A is None else (A, f, gradf, g, proxg, y0)
        out.append(fasta.fasta(*args, verbose=False, tolerance=1e-8, evaluate_objective=True))
    a, b = out
    assert a.iteration_count == b.iteration_count > 3
    assert np.array_equal(a.residuals, b.residuals) and np.array_equal(a.solution, b.solution)
    assert a.solution.min() >= 0 and a.solution.max() <= C
    assert a.objectives[a.iteration_count] < a.objectives[0]


def test_three_argument_linear_operator_alias():
    """`LinearOperator(map, adj, shape)` as democratic_representation.py:80-82 calls it."""
    op = LinearOperator(lambda x: 2 * x, lambda y: 2 * y, (5,))
    assert op.Vshape == op.Wshape == (5,)
    np.random.seed(0)
    c = fasta.fasta(op, lambda z: .5 * la.norm(z - 1) ** 2, lambda z: z - 1, None, None, np.zeros(5), verbose=False)
    np.testing.assert_allclose(c.solution, 0.5 * np.ones(5), atol=1e-4)


def test_backend_switch():
    meta, z = H.load_case("sparse_ls_64x128_adaptive")
    d = H.case_data(meta, z)
    A, At, f, gradf, g, proxg, x0 = reference_style_operands("sparse_ls", d, "as_examples")
    with pytest.raises(TypeError, match="backend='hip'"):
        fasta.fasta(A, At, f, gradf, g, proxg, x0, verbose=False, backend="hip")        # closures cannot run in a kernel
    with pytest.raises(ValueError):
        fasta.fasta(A, At, f, gradf, g, proxg, x0, verbose=False, backend="cuda")
    # tagged operands are ordinary callables with the reference's semantics: backend="numpy" runs them on the host, bitwise
    ls, reg = fasta.LeastSquares(d["b"]), fasta.Shrink(float(d["mu"]))
    np.random.seed(meta["solver_seed"])
    o = H.resolve_options(meta["options"], stopping)
    c = fasta.fasta(A, At, ls.f, ls.gradf, reg.g, reg.prox, x0, verbose=False, backend="numpy", **o)
    assert_bitwise(z, c)
    # a user's own g next to a tagged prox is not something the device loop can honour: generic loop, also bitwise
    np.random.seed(meta["solver_seed"])
    c = fasta.fasta(A, At, ls.f, ls.gradf, g, reg.prox, x0, verbose=False, **o)
    assert_bitwise(z, c)


def test_prox_functions_have_reference_semantics_on_host_arrays(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat_prox.npz"))
    x, xr = k["x"], k["xr"]
    assert np.array_equal(proximal.shrink(x, 1.0), k["shrink_t1"])
    assert np.signbit(proximal.shrink(x, 1.0)[1])                                        # -0.0 for small negatives
    for t, key in ((1.0, "linf_t1"), (4.0, "linf_t4"), (10.5, "linf_t10p5"), (11.0, "linf_t11")):
        assert np.array_equal(proximal.project_Linf_ball(x, t), k[key]), key
    for t, key in ((4.0, "l1_t4"), (1.0, "l1_t1"), (10.5, "l1_t10p5")):
        assert np.array_equal(proximal.project_L1_ball(x, t), k[key]), key
    assert np.array_equal(proximal.shrink(xr, 0.3), k["shrink_r"])
    assert np.array_equal(proximal.project_Linf_ball(xr, 7.0), k["linf_r"])
    assert np.array_equal(proximal.project_L1_ball(xr, 7.0), k["l1_r"])
    assert np.array_equal(x, [3, -1, .5, -4, 0, 2.0])                                    # inputs untouched
    # nuclear-norm prox (fasta/proximal.py:44-55): U diag(shrink(s, t)) V, written without the padded diagonal matrix
    X = np.random.RandomState(1).randn(7, 4)
    U, s, V = la.svd(X)
    S = np.zeros(X.shape)
    S[:len(s), :len(s)] = np.diag(proximal.shrink(s, 0.8))
    np.testing.assert_allclose(proximal.project_Lnuc_ball(X, 0.8), U @ S @ V, rtol=1e-12, atol=1e-14)


def test_config1_examples_package_without_a_gpu(capsys):
    """BASELINE config 1 as written: LASSO 512x1024 on the NumPy CPU path via fasta.examples -- the fixtures
    `c1_sparse_ls_512x1024_*` hold the reference's own run of the same recipe (seed 21 / 201)."""
    from fasta.examples import test_modes
    from fasta.examples.sparse_least_squares import SparseLeastSquaresProblem
    np.random.seed(21)
    problem, x0 = SparseLeastSquaresProblem.construct(M=512, N=1024, K=10, backend="numpy")
    seeds = iter([201, 201, 201])
    solve = problem.solve

    def seeded(x, opts):
        np.random.seed(next(seeds))
        return solve(x, opts)
    problem.solve = seeded
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        runs = test_modes(problem, x0)
    text = capsys.readouterr().out
    assert text.count("Completed in") == 3 and "Computing adaptive FBS." in text
    for (sol, c), mode in zip(runs, ("adaptive", "accelerated", "plain")):
        _, z = H.load_case(f"c1_sparse_ls_512x1024_{mode}")
        assert_bitwise(z, c)
        assert np.array_equal(sol, z["solution"])


def test_example_command_line_runs_without_a_gpu():
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "fasta.examples.sparse_least_squares", "--backend", "numpy"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert res.stdout.count("Completed in") == 3 and "recovery error" in res.stdout


def test_generic_module_never_touches_the_device_or_the_oracle():
    src = open(os.path.join(ROOT, "fasta_python_amd", "generic.py")).read()
    assert "oracle" not in src.replace("the oracle", "") and "hip." not in src and "HipContext" not in src


@pytest.mark.parametrize("name", ["l1ball_64x128_adaptive", "l1ball_64x128_accelerated", "l1ball_64x128_plain", "sparse_ls_64x128_adaptive"])
def test_from_matrix_with_closures_runs_without_a_gpu_and_matches_the_reference_bitwise(name):
    """examples/lasso.py:79 builds its operator with `LinearOperator.from_matrix(A)` and solves with Python closures (:42-47).
    `from_matrix` returns the device-recognisable DenseMatrixMap, but it uploads LAZILY: called on host arrays it applies
    `A @ x` / `A.T @ y` on the host (fasta/linalg.py:41), so this body runs on a GPU-less box, never creates a device context, and
    reproduces the fixture captured from the reference bit for bit."""
    meta, z = H.load_case(name)
    d = H.case_data(meta, z)
    A, b, mu = d["A"], d["b"], float(d["mu"])
    op = LinearOperator.from_matrix(A)                                         # lasso.py:79
    assert op.Vshape == (A.shape[1],) and op.Wshape == (A.shape[0],)
    f = lambda z_: .5 * la.norm((z_ - b).ravel()) ** 2                         # lasso.py:42-45
    gradf = lambda z_: z_ - b
    if meta["kind"] == "l1ball":
        g = lambda x: 0
        proxg = lambda x, t: proximal.project_L1_ball(x, mu)
    else:
        g = lambda x: mu * la.norm(x.ravel(), 1)
        proxg = lambda x, t: proximal.shrink(x, t * mu)
    o = H.resolve_options(meta["options"], stopping)
    np.random.seed(meta["solver_seed"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c = fasta.fasta(op, f, gradf, g, proxg, np.zeros(A.shape[1]), verbose=False, **o)      # lasso.py:47
    assert_bitwise(z, c)
    assert op._ctx is None                                                     # no device context was ever created
    assert np.array_equal(op(np.ones(A.shape[1])), A @ np.ones(A.shape[1])) and np.array_equal(op.H(b), A.T @ b)
    with pytest.raises(AssertionError):
        op(np.zeros(A.shape[1] + 1))                                           # fasta/linalg.py:58
    with pytest.raises(AssertionError):
        LinearOperator.from_matrix(np.zeros(3))                                # fasta/linalg.py:40
    op.close()
