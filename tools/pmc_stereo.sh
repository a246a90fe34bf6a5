export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_LDS -d /tmp/pmc_st -o s --output-format csv -- python $GRAFT_REPO_ROOT/tools/stereo_bench.py > /tmp/pmc.log 2>&1
tail -5 /tmp/pmc.log; find /tmp/pmc_st -type f | head; python - <<"PY"
import csv, glob, collections
f = glob.glob("/tmp/pmc_st/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60] + " grid=" + r["Grid_Size"]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "update_feature" not in k: continue
    w = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
    print(k, "waves %.0f" % w, " ".join("%s/wave=%.0f" % (c, sum(x) / len(x) / w) for c, x in v.items() if c != "SQ_WAVES"))
PY
