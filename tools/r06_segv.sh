#!/bin/bash
# tools/r06_segv.sh -- where a profiled process of this library dies at exit (round-5 verdict, weak #6).  GPU box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/segv
mkdir -p $OUT
CASE="${1:-single:320x240}"
echo "== plain rocprofv3 run"
timeout 200 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python tools/profile_case.py $CASE > $OUT/plain.log 2>&1
echo "exit code $?" | tee -a $OUT/plain.log
tail -5 $OUT/plain.log
echo "== the same under rocgdb"
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt_gdb -o kt --output-format csv -- /opt/rocm/bin/rocgdb -batch -ex "set pagination off" \
   -ex "set confirm off" -ex "handle SIGSEGV stop print" -ex run -ex "bt 40" -ex "info sharedlibrary" -ex "thread apply all bt 16" \
   --args python tools/profile_case.py $CASE > $OUT/gdb.log 2>&1
echo "gdb exit $?"
grep -n "SIGSEGV\|^#" $OUT/gdb.log | head -80
