"""tools/launch_intercept_fit.py KERNEL_TRACE_CSV -- see tools/launch_intercept_trace.py."""
import csv
import sys

import numpy as np

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_persistent") or "k_persistent" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
gap = np.array([(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows[:-1], rows[1:])])
ns = (8, 16, 50, 100, 200, 400, 800)
rows, dur = rows[-12 * len(ns):], dur[-12 * len(ns):]
gap = gap[-12 * len(ns) + 1:]
med = [float(np.median(dur[12 * i + 2:12 * i + 12])) for i in range(len(ns))]
A = np.vstack([np.ones(len(ns)), ns]).T
(b, a), *_ = np.linalg.lstsq(A, np.array(med), rcond=None)
print("kernel duration (rocprofv3) by iterations: " + ", ".join("%d: %.1f us" % (n, m) for n, m in zip(ns, med)))
print("fit: %.2f us + %.4f us per iteration" % (b, a))
inner = [float(np.median(gap[12 * i + 1:12 * i + 11])) for i in range(len(ns))]
print("gap between two launches of the stream (end -> start), median by group: " + ", ".join("%d: %.1f us" % (n, m) for n, m in zip(ns, inner)))
