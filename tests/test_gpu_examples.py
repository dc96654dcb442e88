"""The example harness on the HIP path (BASELINE config 1: 512x1024 sparse least squares through
`fasta.examples`), compared with the oracle on the same seeded instance."""
import warnings

import numpy as np
import pytest

from oracle import fasta_np as fo
from oracle import problems as pr

pytestmark = pytest.mark.gpu


def _oracle_modes(P):
    out = []
    for adaptive, accelerate in ((True, False), (False, True), (False, False)):
        np.random.seed(77)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out.append(fo.fasta(*P.args7(), tolerance=1e-5, evaluate_objective=True, adaptive=adaptive,
                                accelerate=accelerate))
    return out


@pytest.mark.parametrize("devices", [None, [0, 0, 0, 0]])
def test_config1_sparse_least_squares_through_examples_package(capsys, devices):
    """devices=[0]*4: the same example with A row-sharded over four in-process blocks (`--devices 0,0,0,0` on the command line)."""
    import fasta                                        # the drop-in name
    from fasta.examples import test_modes
    from fasta.examples.sparse_least_squares import SparseLeastSquaresProblem
    problem, x0 = SparseLeastSquaresProblem.construct(M=512, N=1024, K=10, seed=21, devices=devices)
    np.random.seed(21)
    P = pr.sparse_least_squares(M=512, N=1024, K=10)
    assert np.array_equal(P.data["b"], problem.b)       # same instance as the reference recipe would draw
    want = _oracle_modes(P)

    class Seeded(type(problem)):
        def solve(self, x0, fasta_options=None):
            np.random.seed(77)
            return super().solve(x0, fasta_options)
    problem.__class__ = Seeded
    got = test_modes(problem, x0)
    problem.close()
    assert "Completed in" in capsys.readouterr().out
    for (sol, c), w in zip(got, want):
        assert isinstance(c, fasta.Convergence)
        assert c.iteration_count == w.iteration_count and c.backtracks == w.backtracks
        k = c.iteration_count
        np.testing.assert_allclose(c.residuals[:k], w.residuals[:k], rtol=1e-6)
        np.testing.assert_allclose(c.objectives[:k + 1], w.objectives[:k + 1], rtol=1e-8)
        np.testing.assert_allclose(sol, w.solution, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("module,cls,kw", [
    ("nn_least_squares", "NNLeastSquaresProblem", dict(M=300, N=200, seed=3)),
    ("lasso", "LASSOProblem", dict(M=100, N=300, seed=4)),
    ("tv_denoising", "TVDenoisingProblem", dict(shape=(64, 80), square=16, seed=5)),
    ("sparse_logistic", "SparseLogisticProblem", dict(M=200, N=300, K=4, mu=8.0, seed=6)),
])
def test_other_examples_match_their_host_twin(module, cls, kw):
    """Each example on the device (backend "hip") against the same instance on the generic host loop (backend "numpy":
    the reference's closures, bit-pinned to the reference by tests/test_generic_cpu.py): identical iteration and backtrack
    counts, histories rtol 1e-6, solution rtol 1e-5."""
    import importlib
    mod = importlib.import_module("fasta_python_amd.examples." + module)
    opts = {"tolerance": 1e-4, "max_iters": 400, "evaluate_objective": True}
    runs = {}
    for backend in ("hip", "numpy"):
        problem, x0 = getattr(mod, cls).construct(backend=backend, **kw)
        try:
            np.random.seed(31)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                runs[backend] = problem.solve(x0, dict(opts))
        finally:
            problem.close()
    (sol, c), (wsol, w) = runs["hip"], runs["numpy"]
    if module == "tv_denoising":
        # adaptive FBS on the TV dual backtracks every few iterations and amplifies last-digit differences (SURVEY.md section 7;
        # the reference-captured fixture tv_32x32_adaptive is pinned on a 40-iteration prefix for the same reason)
        k = 40
        assert min(c.iteration_count, w.iteration_count) >= k
        np.testing.assert_allclose(sol, wsol, atol=2e-2)                 # both reach the same denoised image
    else:
        assert c.iteration_count == w.iteration_count >= 1 and c.backtracks == w.backtracks
        k = c.iteration_count
        np.testing.assert_allclose(sol, wsol, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(c.residuals[:k], w.residuals[:k], rtol=1e-6, atol=1e-13)
    np.testing.assert_allclose(c.stepsizes[:k], w.stepsizes[:k], rtol=1e-6)
    np.testing.assert_allclose(c.objectives[:k + 1], w.objectives[:k + 1], rtol=1e-8)
    assert w.objectives[w.iteration_count] <= w.objectives[0]


def test_verbose_output_format(capsys):
    """fasta/__init__.py:118-120, :302-306 -- header and one line per iteration."""
    import fasta_python_amd as fa
    rng = np.random.RandomState(0)
    A = rng.randn(20, 30) / 10
    ls, reg = fa.LeastSquares(rng.randn(20)), fa.Shrink(0.01)
    np.random.seed(0)
    c = fa.fasta(A, A.T, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(30), backend="hip", max_iters=3, tolerance=0.0)
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "Initializing FASTA..."
    assert out[2] == "Iteration #\tResidual\tStepsize\tAccel. param\tBacktracks\tObjective"
    assert out[3].startswith("[0     ]\t") and len(out[3].split("\t")) == 6
    assert c.iteration_count == 3
