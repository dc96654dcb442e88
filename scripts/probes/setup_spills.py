"""Which instantiations of k_setup_dense (csrc/fh_setup.h) have scratch operations INSIDE a loop?  A scratch reload in the row loop counts
on vmcnt like the row loads, so waiting for it drains the prefetched rows every trip.  Compiles each group to ISA and lists, per kernel,
registers, scratch bytes and the scratch operations inside every loop.  Usage: python scripts/probes/setup_spills.py [group]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(root, "fasta_python_amd", "csrc")
for part in ([int(sys.argv[1])] if len(sys.argv) > 1 else [0, 1]):
    out = os.path.join(tempfile.gettempdir(), f"setup{part}.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", f"-DFH_PART={part}", "-S",
                    "--cuda-device-only", "-o", out, os.path.join(csrc, "fh_setup_part.hip")], check=True, cwd=csrc, stderr=subprocess.DEVNULL)
    L = open(out).read().split("\n")
    starts = [i for i, l in enumerate(L) if re.match(r"^_Z13k_setup_dense\w+:", l)]
    for st in starts:
        end = next(i for i in range(st, len(L)) if "s_endpgm" in L[i])
        name = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", L[st]).groups()
        hdr = {}
        for i in range(st, end):
            m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d)", L[i])
            if m:
                hdr[m.group(1)] = (i, int(m.group(2)))
        sc = [i for i in range(st, end) if "scratch_" in L[i]]
        inside = 0
        for lab, (i, d) in hdr.items():
            back = [j for j in range(i, end) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", L[j])]
            if back and d == 1:
                inside += sum(1 for q in sc if i <= q <= max(back))
        meta = "\n".join(L[end:end + 4000])
        print(f"k_setup_dense<{', '.join(name)}>: scratch ops {len(sc):4d}, inside loops {inside:4d}")
