#!/bin/bash
# HBM read bytes of the one-pass stencil sweep with the plain workgroup order vs ids dealt out XCD by XCD (FH_TUNE_TV_XCD 2 / 1):
# rocprofv3 --pmc FETCH_SIZE over scripts/probes/tune_tvz.py, dispatches split by their order in the run.   bash scripts/probes/tv_xcd_fetch.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_tvxcd
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc" -- python3 scripts/probes/tune_tvz.py 8192 0 0 0 0 2,1 > "$OUT/run.log" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/pmc/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "k_tv_onepass" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
half = len(rows) // 2          # the script runs the reference signature first (xcd = 2), then xcd = 2, then xcd = 1: same number of launches each
P = 8192 * 8192
def summarize(tag, part):
    for name in sorted(set(r["Kernel_Name"] for r in part)):
        v = [2 * float(r["Counter_Value"]) * 1024 for r in part if r["Kernel_Name"] == name]
        print(f"{tag} {name[:44]:44s} launches {len(v):3d}  read {sum(v) / len(v) / 1e9:.4f} GB = {sum(v) / len(v) / P:.2f} B/pixel")
n_ref = 3                       # signature() of the reference configuration: 1 plain + 2 accelerated launches
body = rows[n_ref:]
summarize("plain order  (xcd=2)", body[:len(body) // 2])
summarize("XCD by XCD   (xcd=1)", body[len(body) // 2:])
PY
