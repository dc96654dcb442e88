"""Drop-in name for the reference package: `import fasta; fasta.fasta(...)`, `fasta.linalg`,
`fasta.proximal`, `fasta.stopping` resolve to the MI355X build in `fasta_python_amd`.
(The reference's `fasta.plots` is presentation only and is not part of this build.)"""
import sys as _sys

import fasta_python_amd as _impl
from fasta_python_amd import (EPSILON, Convergence, FBSolver, fasta, linalg, losses, proximal,  # noqa: F401
                              stopping)

__all__ = ["fasta", "Convergence"]

for _name in ("linalg", "proximal", "stopping", "losses", "examples"):
    try:
        _mod = __import__("fasta_python_amd." + _name, fromlist=["_"])
    except ImportError:
        continue
    _sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
