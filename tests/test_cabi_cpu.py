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


def test_struct_layouts_and_constants_of_the_loop_entry_points_match_the_library():
    """fh_run_opts / fh_run_state cross the boundary by pointer: the ctypes mirrors must have the library's sizes (fh_abi_sizes; load_library refuses
    a mismatch), the field order of the header, and the launch-mode / status / tuning constants their header values."""
    import ctypes as C
    lib = hip.load_library()
    sizes = (C.c_uint64 * 4)()
    assert lib.fh_abi_sizes(sizes) == 0
    assert tuple(sizes) == (C.sizeof(hip.RunOpts), C.sizeof(hip.RunState), hip.RUN_HIST, hip.RUN_WINDOW_MAX)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fasta_hip.h")).read(), flags=re.S)
    for struct, mirror in (("fh_run_opts", hip.RunOpts), ("fh_run_state", hip.RunState)):
        body = re.search(r"typedef struct " + struct + r" \{(.*?)\} " + struct + ";", text, flags=re.S).group(1)
        names = [n.split("[")[0] for decl in body.split(";") for n in re.sub(r"^\s*(uint64_t|int64_t|double|int)\s+", "", decl.strip()).replace(" ", "").split(",") if decl.strip()]
        assert names == [f[0] for f in mirror._fields_], struct
    for cname, value in (("FH_LAUNCH_SEPARATE", hip.LAUNCH_SEPARATE), ("FH_LAUNCH_ONEPASS_ALWAYS", hip.LAUNCH_ONEPASS_ALWAYS),
                         ("FH_LAUNCH_ONEPASS_SPECULATIVE", hip.LAUNCH_ONEPASS_SPECULATIVE), ("FH_LAUNCH_PAIR", hip.LAUNCH_PAIR),
                         ("FH_TUNE_RUN_MAX_N", hip.TUNE_RUN_MAX_N), ("FH_TUNE_SEQ_POLL", hip.TUNE_SEQ_POLL), ("FH_TUNE_RUN_CHAIN", hip.TUNE_RUN_CHAIN), ("FH_TUNE_ADJ_CYCLIC", hip.TUNE_ADJ_CYCLIC),
                         ("FH_TUNE_FUSED_CUS", hip.TUNE_FUSED_CUS)):
        m = re.search(cname + r"\s*=\s*(\d+)", text)
        assert m and int(m.group(1)) == value, cname
    for cname, value in (("FH_E_TIMEOUT", hip.E_TIMEOUT), ("FH_E_STATE", hip.E_STATE)):
        m = re.search(r"#define\s+" + cname + r"\s+(\d+)", text)
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


def test_every_row_width_resolves_to_an_instantiated_one_pass_kernel():
    """The one-pass kernel's dispatch is a table (csrc/fh_fused_instances.inc) shared by the instantiations and the host: for EVERY
    row width n = 1 .. 262144, both storages and the A/B variant bits, the shape fused_shape_for() picks must be compiled in --
    there is no default instantiation to fall to.  Beyond 262144 columns there is no one-pass kernel (shape all zeros)."""
    import ctypes as C
    lib = hip.load_library()
    shape = (C.c_int * 5)()
    inst = C.c_int(0)
    seen = set()
    for dtype in (hip.DTYPE_F64, hip.DTYPE_F32_STORAGE):
        for variant in (2, 0, 2 | 8, 2 | 16):
            step = 1 if variant == 2 else 37                  # every n for the default variant, a coprime stride for the A/B bits
            for n in list(range(1, 262145, step)) + [4096, 8192, 16384, 32768, 65536, 131072, 262144]:
                assert lib.fh_fused_shape(n, dtype, variant, 256, shape, C.byref(inst)) == 0
                if dtype == hip.DTYPE_F32_STORAGE and n > 131072:
                    assert tuple(shape) == (0, 0, 0, 0, 0)                  # float32 storage: teams up to 16 members
                    continue
                assert shape[0] > 0 and inst.value == 1, (n, dtype, variant, tuple(shape))
                assert shape[2] * 256 * shape[0] * (4 if dtype else 2) >= n      # the team covers the row
                seen.add((dtype,) + tuple(shape))
    assert lib.fh_fused_shape(262145, hip.DTYPE_F64, 2, 256, shape, C.byref(inst)) == 0 and tuple(shape) == (0, 0, 0, 0, 0)
    assert len(seen) >= 60                                              # ... and nearly the whole table is reachable
