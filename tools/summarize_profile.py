#!/usr/bin/env python3
"""tools/summarize_profile.py ROUND -- turns gpurun_out/prof_<ROUND>/ (tools/profile.sh) into the committed evidence
under profiles/: <ROUND>_kernel_stats_<case>.csv (rocprofv3 --kernel-trace --stats), <ROUND>_counters.json and
profiles/traffic.json (per workload and run path: HBM bytes and VALU instructions per launch, which bench.py reports as
roofline.traffic and as the VALU roofline of the batched lines).

HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM) -> doubled as the guide prescribes.  That correction is
calibrated for wide streaming reads; the persistent kernels' 16-byte sc1 record reads are a different access pattern, so
their absolute figure is an upper bound."""
import csv
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r04"
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
                return {"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                        "max_ns": float(r["MaxNs"])}
    return None


def last_json(path):
    try:
        for line in reversed(open(path).read().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    except Exception:  # noqa: BLE001
        pass
    return None


# tag -> (dominant kernel, traffic.json key or None)
CASES = {"bench": ("k_persistent_pv", "640x480:persistent-pv"), "bench_tv": ("k_persistent_tv", "640x480:persistent-tv"),
         "bench_step": ("k_fused_step", "640x480:per-step hipGraph"), "cfg3": ("k_persistent_pv", "1280x720:persistent-pv"),
         "cfg5": ("k_persistent_pv2", "1920x1080:persistent-pv2"), "batch5": ("k_persistent_pv2", "640x480x5:persistent-pv2"), "batch10": ("k_persistent_pv2", "640x480x10:persistent-pv2"), "batch30": ("k_persistent_tv", "640x480x30:persistent-tv"),
         "batch64": ("k_persistent_tv", "640x480x64:persistent-tv"), "stream64": ("k_fused_step", None),
         "stereo": ("k_update_feature_idepths", None)}
out, traffic = {}, {}
for tag, (kernel, key) in CASES.items():
    kt = os.path.join(SRC, f"kt_{tag}", "kt_kernel_stats.csv")
    if not os.path.exists(kt):
        continue
    shutil.copy(kt, os.path.join(DST, f"{R}_kernel_stats_{tag}.csv"))
    entry = {"kernel": kernel, "kernel_stats": kstat(kt, kernel), "workload": last_json(os.path.join(SRC, f"kt_{tag}.log"))}
    if entry["workload"] and "roofline" in entry["workload"]:  # a bench line: keep the essentials only
        w = entry["workload"]
        entry["workload"] = {"value": w["value"], "ms_per_step": w["ms_per_step"], "run_path": w.get("run_path"), "roofline": w["roofline"]}
    c = {}
    for d, f in ((f"fetch_{tag}", "f_counter_collection.csv"), (f"write_{tag}", "w_counter_collection.csv"), (f"sq_{tag}", "s_counter_collection.csv"),
                 (f"lds_{tag}", "l_counter_collection.csv"), (f"act_{tag}", "a_counter_collection.csv")):
        p = os.path.join(SRC, d, f)
        if os.path.exists(p):
            c.update(counters(p, kernel))
    entry["counters_per_launch"] = c
    t = {"kernel": kernel, "source": f"profiles/{R}_counters.json"}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        hbm = (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0
        entry["hbm_bytes_per_launch"] = hbm
        entry["hbm_formula"] = "(2*FETCH_SIZE + WRITE_SIZE)*1024  [KiB counters; x2 = gfx950 FETCH_SIZE correction]"
        t["hbm_bytes_per_launch"] = round(hbm)
        if entry["kernel_stats"]:
            entry["measured_hbm_GBps"] = round(hbm / (entry["kernel_stats"]["avg_ns"] * 1e-9) / 1e9, 1)
    if "SQ_INSTS_VALU" in c:
        t["valu_insts_per_launch"] = round(c["SQ_INSTS_VALU"]["mean"])
        if entry["kernel_stats"]:  # chip VALU issue peak: 1024 SIMDs x one wave-instruction per 2 cycles at 2.4 GHz
            rate = c["SQ_INSTS_VALU"]["mean"] / (entry["kernel_stats"]["avg_ns"] * 1e-9)
            entry["valu_insts_per_s"] = rate
            entry["valu_issue_frac_of_peak"] = round(rate / (1024 * 1.2e9), 4)
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]["mean"] > 0:
        entry["wait_any_over_wave_cycles"] = round(c["SQ_WAIT_ANY"]["mean"] / c["SQ_WAVE_CYCLES"]["mean"], 4)
        if "SQ_ACTIVE_INST_ANY" in c:
            entry["active_inst_over_wave_cycles"] = round(c["SQ_ACTIVE_INST_ANY"]["mean"] / c["SQ_WAVE_CYCLES"]["mean"], 4)
    if "SQ_ACTIVE_INST_LDS" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]["mean"] > 0:
        # the on-chip side of a resident kernel (tools/profile.sh prof_onchip): per wave-cycle, how much of the time an instruction of
        # each class is in flight, and the LDS pipe's own counters.  SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles
        # summed over waves; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE count LDS-pipe cycles (conflict share = their ratio).
        wc = c["SQ_WAVE_CYCLES"]["mean"]
        oc = {k.lower().replace("sq_", "") + "_over_wave_cycles": round(c[k]["mean"] / wc, 4)
              for k in ("SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA") if k in c}
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", {}).get("mean", 0) > 0:
            oc["lds_bank_conflict_share_of_lds_active"] = round(c["SQ_LDS_BANK_CONFLICT"]["mean"] / c["SQ_LDS_IDX_ACTIVE"]["mean"], 4)
        if entry["kernel_stats"]:
            secs = entry["kernel_stats"]["avg_ns"] * 1e-9
            for k in ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU"):
                if k in c:
                    oc[k.lower().replace("sq_", "") + "_per_s"] = c[k]["mean"] / secs
            if "SQ_LDS_IDX_ACTIVE" in c:  # LDS pipe busy: cycles the LDS index stage is active, per CU-cycle available (256 CUs x 2.4 GHz)
                oc["lds_pipe_busy_frac"] = round(c["SQ_LDS_IDX_ACTIVE"]["mean"] / (secs * 256 * 2.4e9), 4)
        entry["on_chip"] = oc
        t["on_chip"] = oc
    if key:
        traffic[key] = t
    out[tag] = entry
# the feature-update kernel: instructions and (quad-)cycles per wave, by template instance (lanes per feature) and grid
sq = os.path.join(SRC, "sq_stereo", "s_counter_collection.csv")
if os.path.exists(sq):
    agg = {}
    with open(sq) as f:
        for r in csv.DictReader(f):
            if "k_update_feature_idepths" not in r["Kernel_Name"]:
                continue
            lanes = "16" if "<16>" in r["Kernel_Name"] or "ILi16E" in r["Kernel_Name"] else "1"
            agg.setdefault((lanes, int(r["Grid_Size"])), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    rows = []
    for (lanes, grid), v in sorted(agg.items(), key=lambda kv: (kv[0][1] // int(kv[0][0]), kv[0][0])):
        w = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
        row = {"lanes_per_feature": int(lanes), "features": grid // int(lanes), "waves": round(w)}
        for c, x in v.items():
            if c != "SQ_WAVES":
                row[c + "_per_wave"] = round(sum(x) / len(x) / w, 1)
        row["note"] = "SQ_WAVE_CYCLES counts quad-cycles"
        rows.append(row)
    out.setdefault("stereo", {"kernel": "k_update_feature_idepths"})["per_wave_counters"] = rows
    log = os.path.join(SRC, "kt_stereo.log")
    if os.path.exists(log):
        out["stereo"]["bench_lines"] = [ln.strip() for ln in open(log).read().splitlines() if " feats " in ln]
json.dump(out, open(os.path.join(DST, f"{R}_counters.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: v.get(kk) for kk in ("kernel_stats", "hbm_bytes_per_launch", "valu_issue_frac_of_peak", "wait_any_over_wave_cycles")} for k, v in out.items()}, indent=1)[:6000])
