#!/bin/bash
# SQ / TCC counter passes over the TV bench (what the one-pass sweep's waves spend their cycles on).  Usage: bash scripts/probes/profile_tv_sq.sh <tag>
set -u
TAG=${1:-r03tvsq}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" \
           "TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_32B_sum TCC_TAG_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc_$i" -- python3 bench.py --workload tv --steps 6 --warmup 1 ${EXTRA:-} > "$OUT/bench_pmc_$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_tv_onepass" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in agg.items():
        print(k, file=fh)
        for name, v in sorted(d.items()):
            print(f"   {name:36s} mean {sum(v) / len(v):16.1f}  (n={len(v)})", file=fh)
print(open(out + "/summary.txt").read())
PY
