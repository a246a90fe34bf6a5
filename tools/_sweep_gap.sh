#!/bin/bash
export FLAME_NLTGV2_DEBUG_PV_CAP=28 FLAME_NLTGV2_DEBUG_ROWPACK_MAX=40
for b in 0 1 3 7; do
  echo "== GAP $((b+1)) x s_sleep 1"
  FLAME_NLTGV2_DEBUG_BETA=$b PV_VARIANTS="13=4,8=9;13=4,8=13;13=4,8=17;13=4,8=21" timeout 120 python tools/pv_big.py "$@" 2>&1 | grep -v "upload_graph\|amdgpu.ids\|pv: " | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['case'], {k:(v['us_per_iter_mean'],v['path'][-3:]) for k,v in d.items() if isinstance(v,dict) and k not in ('auto','he','pv')})
"
done
