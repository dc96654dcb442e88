"""Drop-in name for the reference package: `import fasta; fasta.fasta(...)`, `fasta.linalg`,
`fasta.proximal`, `fasta.stopping`, `fasta.examples` resolve to the MI355X build in `fasta_python_amd`.
(The reference's `fasta.plots` is presentation only and is not part of this build.)"""
import sys as _sys

import fasta_python_amd as _impl
from fasta_python_amd import *          # noqa: F401,F403  (fasta, Convergence, tagged operands, ...)
from fasta_python_amd import EPSILON, hip, linalg, losses, proximal, stopping      # noqa: F401

__all__ = list(_impl.__all__)

for _name in ("linalg", "proximal", "stopping", "losses", "hip", "synthetic", "examples"):
    _mod = __import__("fasta_python_amd." + _name, fromlist=["_"])
    _sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
