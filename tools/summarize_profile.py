#!/usr/bin/env python3
"""tools/summarize_profile.py ROUND -- turns gpurun_out/prof_<ROUND>/ (tools/profile.sh) into the
committed evidence under profiles/: <ROUND>_kernel_stats*.csv (rocprofv3 --kernel-trace --stats),
<ROUND>_counters.json and profiles/traffic.json (HBM bytes per launch that bench.py reports as
roofline.traffic).

HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM) -> doubled as the guide
prescribes.  That correction is calibrated for wide streaming reads; the persistent kernel's 16-byte
sc1 record reads are a different access pattern, so its absolute figure is an upper bound."""
import csv
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def counters(path, kernel):
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {"mean": sum(v) / len(v), "n": len(v)} for k, v in agg.items()}


def kstat(path, kernel):
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Name"]:
                return {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                        "max_ns": float(r["MaxNs"])}
    return None


out = {}
traffic = {}
for tag, sub, kernel, path_name in (("persistent", "", "k_persistent_he", "persistent"),
                                    ("per_step", "_step", "k_fused_step", "per-step hipGraph"),
                                    ("stream_64_frames", "_stream", "k_fused_step", None),
                                    ("resident_30_frames", "_batch", "k_persistent_tv", None),
                                    ("feature_update", "_stereo", "k_update_feature_idepths", None)):
    kt = os.path.join(SRC, "kt" + sub, "kt_kernel_stats.csv")
    if not os.path.exists(kt):
        continue
    shutil.copy(kt, os.path.join(DST, f"{R}_kernel_stats{sub}.csv"))
    entry = {"kernel": kernel, "kernel_stats": kstat(kt, kernel)}
    c = {}
    for d, f in (("fetch" + sub, "f_counter_collection.csv"), ("write" + sub, "w_counter_collection.csv"),
                 ("sq" + sub, "s_counter_collection.csv")):
        p = os.path.join(SRC, d, f)
        if os.path.exists(p):
            c.update(counters(p, kernel))
    entry["counters_per_launch"] = c
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        hbm = (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0
        entry["hbm_bytes_per_launch"] = hbm
        entry["hbm_formula"] = "(2*FETCH_SIZE + WRITE_SIZE)*1024  [KiB counters; x2 = gfx950 FETCH_SIZE correction]"
        if path_name:
            traffic[f"640x480:{path_name}"] = {"hbm_bytes_per_launch": round(hbm), "kernel": kernel,
                                                "source": f"profiles/{R}_counters.json"}
        if entry["kernel_stats"]:
            entry["measured_hbm_GBps"] = round(hbm / (entry["kernel_stats"]["avg_ns"] * 1e-9) / 1e9, 1)
    out[tag] = entry
json.dump(out, open(os.path.join(DST, f"{R}_counters.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
