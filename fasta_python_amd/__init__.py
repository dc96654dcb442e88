"""fasta_python_amd -- MI355X-native forward-backward splitting behind the fasta-python call surface.

    from fasta_python_amd import fasta, Convergence, linalg, proximal, stopping, losses

`fasta(A[, At], f, gradf, g, proxg, x0, ...)` keeps the reference's signature, options, defaults
and `Convergence` record (reference: fasta/__init__.py:38-53, :323-351); the work runs in
hand-written HIP kernels through the ctypes C ABI of include/fasta_hip.h.  The top-level `fasta`
package in this repository re-exports these names so `import fasta` keeps working.

Which loop runs is decided by the operand types: device-recognisable operands (matrix / DenseMatrixMap / GradDivMap + tagged
loss + tagged prox) run the HIP loop and raise when the built `libfasta_hip.so` or a gfx950 GPU is missing -- no CPU fallback;
closures, callable pairs, `A=None` and host LinearMaps cannot execute inside a kernel and run the generic host loop
(`generic.py`, the reference's semantics).  `backend="hip"` / `backend="numpy"` force either.
"""

from . import generic, hip, linalg, losses, proximal, stopping
from .linalg import DenseMatrixMap, GradDivMap, LinearMap, LinearOperator, ShardedDenseMatrixMap
from .losses import LeastSquares, LogisticLoss
from .proximal import Box, L1Ball, LinfProx, NonNeg, NoProx, Shrink, TVDualBall
from .solver import EPSILON, Convergence, FBSolver, fasta

__all__ = ["fasta", "Convergence", "FBSolver", "EPSILON", "linalg", "proximal", "stopping", "losses", "hip",
           "LinearMap", "LinearOperator", "DenseMatrixMap", "ShardedDenseMatrixMap", "GradDivMap", "LeastSquares", "LogisticLoss",
           "Shrink", "NonNeg", "LinfProx", "L1Ball", "Box", "TVDualBall", "NoProx"]
