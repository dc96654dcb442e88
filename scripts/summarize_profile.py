"""Condense gpurun_out/prof_<tag>/ (rocprofv3 csv output of scripts/profile_bench.sh) into the small,
tracked files under profiles/:  <tag>_kernel_stats.csv  and  <tag>_pmc_summary.json.

HBM traffic per launch follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE counts exactly half the bytes of a 16-B-per-lane
streaming read -> read bytes = 2 * FETCH_SIZE * 1024 (calibrated here against k_stream_read, which reads
a known byte count with the same access width); WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))


def mean_counter(kind, counter):
    f = glob.glob(os.path.join(src, f"pmc_{kind}", "*", "*_counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch = mean_counter("fetch", "FETCH_SIZE")
write = mean_counter("write", "WRITE_SIZE")
out = {"units": "bytes per launch", "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); "
       "read = 2*FETCH_SIZE*1024 (gfx950 correction), write = WRITE_SIZE*1024", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not any(s in k for s in ("k_fwd", "k_adj", "k_fused", "k_stream", "k_tv", "k_gen", "k_setup", "k_run", "k_level")):
        continue
    rd = 2 * fetch.get(k, (0, 0))[0] * 1024
    wr = write.get(k, (0, 0))[0] * 1024
    out["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr,
                         "launches_sampled": fetch.get(k, (0, 0))[1]}
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
