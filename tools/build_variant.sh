#!/bin/bash
# tools/build_variant.sh NAME "-DFLAGS" -- experimental build of the library with other flags for nltgv2_persistent.hip (build/ab/libNAME.so)
set -e
R=/root/repo; FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$R/include -I$R/flame_amd/csrc -Wall -Wno-unused-result"
mkdir -p $R/build/obj $R/build/ab
/opt/rocm/bin/hipcc $FL $2 -c $R/flame_amd/csrc/nltgv2_persistent.hip -o $R/build/obj/pers_$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build/ab/lib$1.so $R/build/obj/pers_$1.o $R/build/obj/nltgv2_*.o $R/build/obj/stereo_*.o $R/build/obj/frames_capi.o $R/build/obj/delaunay.o -Wl,-rpath,/opt/rocm/lib -ldl
