"""Scratch operations INSIDE loops, per kernel, for a source file of csrc/ (a scratch reload in a streaming loop counts on vmcnt like
the row loads: waiting for it drains the prefetched rows).  Usage: python scripts/loop_spills.py <file.hip> <kernel-name-regex> [-DFLAG ...]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "fasta_python_amd", "csrc")
src, pat, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
out = os.path.join(tempfile.gettempdir(), f"loop_spills_{os.getpid()}.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-S", "--cuda-device-only", "-o", out,
                os.path.join(csrc, src)] + flags, check=True, cwd=csrc, stderr=subprocess.DEVNULL)
L = open(out).read().split("\n")
starts = [i for i, l in enumerate(L) if re.match(r"^_Z\w+:", l) and re.search(pat, l)]
for st in starts:
    end = next(i for i in range(st, len(L)) if "s_endpgm" in L[i])
    hdr = {}
    for i in range(st, end):
        m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d)", L[i])
        if m:
            hdr[m.group(1)] = (i, int(m.group(2)))
    sc = [i for i in range(st, end) if "scratch_" in L[i]]
    rows = []
    for lab, (i, d) in hdr.items():
        back = [j for j in range(i, end) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", L[j])]
        if back:
            n = sum(1 for q in sc if i <= q <= max(back))
            if n:
                rows.append(f"depth {d}: {n}")
    print(f"{L[st][:60]:60s} scratch ops {len(sc):4d}; inside loops: {', '.join(rows) if rows else 'none'}")
os.remove(out)
