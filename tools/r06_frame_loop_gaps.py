"""tools/r06_frame_loop_gaps.py TRACE.csv -- what the device does between two solver launches of the C++ frame loop (a rocprofv3
kernel trace of tests/cpp/frame_loop_test.cc in lean mode): the share of the time the solver's kernels run, the gaps between them by
length, and which kernels fill the gaps."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))


def short(name):
    for cut in ("flame_hip::", "(anonymous namespace)::", "void ", "rocprim::ROCPRIM_400200_NS::detail::"):
        name = name.replace(cut, "")
    return name.split("(")[0].split("<")[0][:40]


ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows))
solver = [e for e in ev if "k_persistent" in e[2] or "k_fused_step" in e[2]]
t0, t1 = solver[0][0], solver[-1][1]
busy = sum(e[1] - e[0] for e in solver)
print(f"{len(ev)} kernels, {len(solver)} solver launches over {(t1 - t0) / 1e6:.2f} ms; solver kernels run {100.0 * busy / (t1 - t0):.1f} % of that time "
      f"(mean launch {busy / len(solver) / 1e3:.1f} us)")
gaps = []
for a, b in zip(solver, solver[1:]):
    g = b[0] - a[1]
    inside = [e for e in ev if e[0] >= a[1] and e[1] <= b[0] and e not in (a, b)]
    gaps.append((g, inside))
tot = sum(g for g, _ in gaps)
for lo, hi in ((0, 5e3), (5e3, 20e3), (20e3, 60e3), (60e3, 150e3), (150e3, 1e12)):
    sel = [(g, i) for g, i in gaps if lo <= g < hi]
    if sel:
        print(f"  gaps {lo / 1e3:6.0f}..{hi / 1e3:6.0f} us: {len(sel):4d}, {sum(g for g, _ in sel) / 1e6:7.3f} ms = {100.0 * sum(g for g, _ in sel) / (t1 - t0):5.1f} % of the time; "
              f"kernels inside cover {100.0 * sum(e[1] - e[0] for _, i in sel for e in i) / max(1, sum(g for g, _ in sel)):.0f} % of these gaps")
by = defaultdict(lambda: [0, 0])
for g, inside in gaps:
    for e in inside:
        by[e[2]][0] += 1
        by[e[2]][1] += e[1] - e[0]
print("  kernels inside the gaps (count, total us):")
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"    {n:5d} {t / 1e3:9.1f}  {k}")
big = sorted(range(len(gaps)), key=lambda i: -gaps[i][0])
# the timeline around the third-largest gap: every kernel from the solver launch before it to the one after it (start offset us, duration us, queue)
i = big[min(2, len(big) - 1)]
a, b = solver[i], solver[i + 1]
print(f"  timeline of one frame's largest gap ({gaps[i][0] / 1e3:.1f} us), from the end of the solver launch before it (t = 0):")
for e in ev:
    if e[0] >= a[0] and e[1] <= b[1]:
        print(f"    t={(e[0] - a[1]) / 1e3:9.1f} us  {((e[1] - e[0]) / 1e3):7.1f} us  q{e[3]}  {e[2]}")
# ... and the whole of that frame: every gap > 20 us with what stands in it
lo = solver[max(0, i - 8)][0]
hi = solver[min(len(solver) - 1, i + 8)][1]
print("  the launches around it (solver launches as S<us>, gaps as [us: kernels inside]):")
line = []
for j in range(max(0, i - 8), min(len(solver) - 1, i + 8)):
    line.append(f"S{(solver[j][1] - solver[j][0]) / 1e3:.0f}")
    g, inside = gaps[j]
    line.append(f"[{g / 1e3:.0f}: {len(inside)}]")
print("    " + " ".join(line))
