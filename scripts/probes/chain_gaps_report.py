"""Gaps between consecutive k_fused_* dispatches in a rocprofv3 kernel trace: python scripts/probes/chain_gaps_report.py <dir>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
dur, gap = (en - st) * 1e-3, (st[1:] - en[:-1]) * 1e-3
gap = gap[20:]
print(f"{rows[-1]['Kernel_Name'][:60]}: {len(rows)} launches; duration median {np.median(dur[20:]):.2f} us; gap to the next launch: median {np.median(gap):.2f} us, "
      f"p10 {np.percentile(gap, 10):.2f}, p90 {np.percentile(gap, 90):.2f}; start-to-start median {np.median(np.diff(st)[20:]) * 1e-3:.2f} us")
