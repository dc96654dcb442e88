"""hipcc's register allocation for the one-wave-per-SIMD streaming kernels is at the edge of the register file: an unrelated edit (round 5: removing an
unused pointer from the kernel's parameter struct) can move one spill into a row loop, where its reload -- counted on vmcnt like the row loads -- drains the
prefetched rows on every trip (+14 % at 32768^2, silently: every parity test still passes).  This test compiles the default shapes of the headline sizes to ISA
and fails on any scratch operation inside a loop.  The whole library is checked by scripts/check_loop_spills.sh (profiles/r05_loop_spills.txt)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fasta_python_amd", "csrc")

SHAPES = """#include "fh_fused.h"
#include "fh_setup.h"
template __global__ void k_fused_dense<8, 1, 2, 16, 0, 0, 0>(const FusedP);      // 65536 columns: the headline
template __global__ void k_fused_dense<8, 1, 1, 8, 0, 0, 0>(const FusedP);       // 32768
template __global__ void k_fused_dense<8, 1, 1, 4, 0, 0, 0>(const FusedP);       // 16384
template __global__ void k_fused_dense<8, 1, 1, 2, 0, 0, 0>(const FusedP);       // 8192
template __global__ void k_fused_dense<8, 1, 1, 1, 0, 0, 0>(const FusedP);       // 4096
template __global__ void k_fused_dense<16, 1, 1, 16, 1, 3, 0>(const FusedP);     // 131072 (x slice in LDS)
template __global__ void k_fused_dense<4, 1, 1, 16, 0, 4, 1>(const FusedP);      // 65536, float32 storage, two workgroups per CU
template __global__ void k_setup_dense<8, 2, 16, 512, 2>(const SetupP);          // the set-up at 65536 columns
"""


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_scratch_operation_inside_a_loop_of_the_default_shapes(tmp_path):
    src = os.path.join(CSRC, "_spill_check.hip")
    with open(src, "w") as fh:
        fh.write(SHAPES)
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "loop_spills.py"), "_spill_check.hip", "k_fused_dense|k_setup_dense"],
                             capture_output=True, text=True, timeout=900)
    finally:
        os.remove(src)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 8, out.stdout
    bad = [l for l in lines if "inside loops: none" not in l]
    assert not bad, "scratch operations inside a loop:\n" + "\n".join(bad)
