#!/usr/bin/env python3
"""tools/kernel_stats.py DIR [substring ...] -- per-kernel averages from a rocprofv3 --kernel-trace --stats run (fl_kernel_stats.csv / kt_kernel_stats.csv under
DIR), optionally only the kernels whose name contains one of the substrings; last line: the sum of (total / calls of the most frequent selected kernel)."""
import csv
import glob
import sys

d = sys.argv[1]
want = sys.argv[2:]
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
sel = [r for r in rows if not want or any(w in r["Name"] for w in want)]
per = max(int(r["Calls"]) for r in sel if "feat_build" not in r["Name"] and "calibrate" not in r["Name"]) if sel else 1
base = min(int(r["Calls"]) for r in sel if int(r["Calls"]) > 1) if sel else 1
tot = 0.0
for r in sel:
    n = r["Name"].replace("flame_hip::(anonymous namespace)::", "").replace("void ", "")[:64]
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    print(f"  {n:64s} calls={int(r['Calls']):5d} avg_us={float(r['AverageNs']) / 1e3:9.2f}")
print(f"  sum of the selected kernels: {tot:.1f} us over the run, {tot / base:.1f} us per {base} calls")
