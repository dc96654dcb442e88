"""Working counterparts of the reference's example harness (fasta/examples/__init__.py:19-91).  The reference's own
example modules do not import; these keep their recipes -- same RNG draw order in `construct()`,
same closures in `solve()`, same three-mode driver -- plus an optional `seed=` and no plotting dependency.

Every problem takes `backend=`:
    "hip"    device-tagged operands, the fused MI355X loop (raises without libfasta_hip.so / a GPU);
    "numpy"  the reference's closures as written there, on the generic host loop (BASELINE config 1: no GPU needed).

    python -m fasta.examples.sparse_least_squares [--backend hip|numpy] [--devices 0,1,2,3]
(--devices: sparse_least_squares only -- A row-sharded over these devices of the one process; a repeated id puts every block on that GPU)
(without --backend the command line uses "hip" when a GPU is visible and says so when it is not).
"""

import sys
from abc import ABC, abstractmethod

from .. import Convergence

__all__ = ["ExampleProblem", "print_info", "test_modes", "cli_backend", "cli_devices", "TOLERANCE"]

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

    backend = "hip"
    _device_op = None

    def device_operator(self, make):
        """The problem's operator in HBM, created on first use (backend "hip") and kept for later solves."""
        if self._device_op is None:
            self._device_op = make()
        return self._device_op

    def close(self):
        if self._device_op is not None:
            self._device_op.close()
            self._device_op = None
        A = getattr(self, "A", None)
        if hasattr(A, "close"):
            A.close()


def cli_backend(argv=None):
    """`--backend hip|numpy` of the example command lines.  Default: "hip" when a GPU is visible; otherwise "numpy",
    announced -- an explicit choice of the harness, `fasta()` itself never falls back."""
    argv = sys.argv[1:] if argv is None else argv
    if "--backend" in argv:
        choice = argv[argv.index("--backend") + 1]
        if choice not in ("hip", "numpy"):
            raise SystemExit("--backend must be hip or numpy")
        return choice
    from .. import hip
    try:
        if hip.device_count() > 0:
            return "hip"
    except Exception:
        pass
    print("no MI355X visible: running the generic NumPy host loop (--backend numpy)")
    return "numpy"


def cli_devices(argv=None):
    """`--devices 0,1,...` of the example command lines: row blocks of A over several devices (None when absent)."""
    argv = sys.argv[1:] if argv is None else argv
    if "--devices" not in argv:
        return None
    return [int(d) for d in argv[argv.index("--devices") + 1].split(",")]


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
