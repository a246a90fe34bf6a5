#!/bin/bash
# tools/r06_segv_confirm.sh -- the exit crash under rocprofv3 without this library, and the library with plain first launches.  GPU box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/segv_confirm
mkdir -p $OUT
[ -x tools/coop_exit_repro ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -o tools/coop_exit_repro tools/coop_exit_repro.hip 2>/dev/null
tools/coop_exit_repro coop > $OUT/bare_coop.log 2>&1; echo "bare, cooperative launch:               exit $?"
timeout 200 rocprofv3 --kernel-trace -d $OUT/p -o kt --output-format csv -- tools/coop_exit_repro plain > $OUT/prof_plain.log 2>&1; echo "rocprofv3, plain launch:                exit $?"
timeout 200 rocprofv3 --kernel-trace -d $OUT/c -o kt --output-format csv -- tools/coop_exit_repro coop > $OUT/prof_coop.log 2>&1; echo "rocprofv3, cooperative launch:          exit $?  ($(grep -c 'SIGSEGV' $OUT/prof_coop.log) SIGSEGV line)"
timeout 200 rocprofv3 --kernel-trace -d $OUT/l -o kt --output-format csv -- python tools/profile_case.py single:320x240 > $OUT/lib_default.log 2>&1; echo "rocprofv3, this library (default):      exit $?"
FLAME_NLTGV2_COOPERATIVE=1 timeout 200 rocprofv3 --kernel-trace -d $OUT/lc -o kt --output-format csv -- python tools/profile_case.py single:320x240 > $OUT/lib_coop.log 2>&1; echo "rocprofv3, this library, coop forced:   exit $?"
timeout 200 rocprofv3 --pmc SQ_WAVES -d $OUT/pmc -o s --output-format csv -- python tools/profile_case.py single:320x240 > $OUT/lib_pmc.log 2>&1; echo "rocprofv3 --pmc, this library:          exit $?"
