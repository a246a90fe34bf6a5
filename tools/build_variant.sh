#!/bin/bash
# tools/build_variant.sh NAME "-DFLAGS" -- an experimental build of the library next to the shipped one: nltgv2_persistent.hip is
# compiled with the extra flags, everything else as shipped (objects cached under build/obj, rebuilt when a source is newer),
# linked into build/ab/libNAME.so.  Run a tool against it with FLAME_AMD_LIBRARY=$PWD/build/ab/libNAME.so (build/ travels to the
# GPU box with gpurun; it is git-ignored).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/flame_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$R/include -I$S -Wall -Wno-unused-result"
mkdir -p $R/build/obj $R/build/ab
OBJS=""
for f in nltgv2_kernels.hip nltgv2_persistent_tv.hip nltgv2_layout.hip nltgv2_context.hip nltgv2_run.hip nltgv2_graph_capi.hip nltgv2_frame_capi.hip \
         stereo_kernels.hip stereo_capi.hip frames_capi.hip delaunay.cpp; do
  o=$R/build/obj/${f%.*}.o
  if [ ! -f $o ] || [ -n "$(find $S $R/include -newer $o \( -name '*.hip' -o -name '*.hpp' -o -name '*.h' -o -name '*.cpp' \) | head -1)" ]; then
    /opt/rocm/bin/hipcc $FL -c $S/$f -o $o &
  fi
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc $FL $2 -c $S/nltgv2_persistent.hip -o $R/build/obj/pers_$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build/ab/lib$1.so $R/build/obj/pers_$1.o $OBJS -Wl,-rpath,/opt/rocm/lib -ldl
echo $R/build/ab/lib$1.so
