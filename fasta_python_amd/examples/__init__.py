"""Working counterparts of the reference's example harness (fasta/examples/__init__.py:19-91), running on
the MI355X path.  The reference's own example modules do not import; these keep
their recipes -- same RNG draw order in `construct()`, same closures in `solve()`, same three-mode driver --
with device-tagged operands, an optional `seed=` for reproducibility and no plotting dependency.

    python -m fasta_python_amd.examples.sparse_least_squares
"""

from abc import ABC, abstractmethod

from .. import Convergence

__all__ = ["ExampleProblem", "print_info", "test_modes", "TOLERANCE"]

TOLERANCE = 1E-5          # fasta/examples/__init__.py:16


class ExampleProblem(ABC):
    """Common interface of the example problems (fasta/examples/__init__.py:19-51)."""

    @abstractmethod
    def solve(self, initial_guess, fasta_options=None):
        """Return (solution, Convergence)."""

    @staticmethod
    @abstractmethod
    def construct():
        """Return (problem, initial guess)."""

    def plot(self, solution):
        """Presentation is out of scope for this build (reference: fasta/plots.py)."""

    def close(self):
        A = getattr(self, "A", None)
        if hasattr(A, "close"):
            A.close()


def print_info(solution: Convergence) -> None:
    """fasta/examples/__init__.py:54-60."""
    print("Completed in {} iterations, {:f} seconds.".format(
        solution.iteration_count, solution.times[solution.iteration_count] - solution.times[0]))


def test_modes(problem, x0, extra_options=None):
    """Adaptive, accelerated and plain FBS at TOLERANCE with objective tracking
    (fasta/examples/__init__.py:63-91).  Returns the three (solution, Convergence) pairs."""
    out = []
    for label, adaptive, accelerate in (("adaptive", True, False), ("accelerated", False, True), ("plain", False, False)):
        print()
        print("Computing {} FBS.".format(label))
        opts = {'tolerance': TOLERANCE, 'evaluate_objective': True, 'adaptive': adaptive, 'accelerate': accelerate}
        opts.update(extra_options or {})
        res = problem.solve(x0, opts)
        print_info(res[1])
        out.append(res)
    print()
    return tuple(out)


test_modes.__test__ = False      # not a pytest test
