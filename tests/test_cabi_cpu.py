"""CPU-side checks of the C-ABI boundary: the shared library loads, exports every symbol the header
declares (and nothing the binding does not know), and the host-side operand recognition fails loudly.
No compute call is made (there is no GPU in the build container)."""
import os
import re

import numpy as np
import pytest

from fasta_python_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "fasta_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hip.load_library()
    declared = header_functions()
    assert declared == sorted(hip.SIGNATURES), "binding and header disagree"
    for name in declared:
        assert hasattr(lib, name), name


def test_enums_match_header():
    text = open(os.path.join(ROOT, "include", "fasta_hip.h")).read()
    for cname, value in (("FH_PROX_SHRINK", hip.PROX_SHRINK), ("FH_PROX_BOX", hip.PROX_BOX),
                         ("FH_VEC_T3", hip.VEC_T3), ("FH_S_GMAX_ADJ", hip.S_GMAX_ADJ), ("FH_S_ALPHA", hip.S_ALPHA),
                         ("FH_NSCALARS", hip.NSCALARS), ("FH_TUNE_NT_LOADS", hip.TUNE_NT_LOADS)):
        m = re.search(cname + r"\s*=\s*(\d+)", text)
        assert m and int(m.group(1)) == value, cname


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(hip.HipError):
        hip.load_library(str(tmp_path / "nope.so"))


def test_no_gpu_is_an_error_not_a_fallback():
    import fasta_python_amd as fa
    try:
        n = hip.device_count()
    except hip.HipError:
        n = 0
    if n:
        pytest.skip("a GPU is present")
    ls, reg = fa.LeastSquares(np.zeros(3)), fa.Shrink(0.1)
    with pytest.raises(hip.HipError):
        fa.fasta(np.eye(3), np.eye(3), ls.f, ls.gradf, reg.g, reg.prox, np.zeros(3), verbose=False)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "fasta_python_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
