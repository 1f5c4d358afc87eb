#!/bin/bash
# tools/build_variant.sh NAME "-DMACRO=.. ..."  -> build_ab/libNAME.so: the library with rg_mp3dev.hip / rg_k2_tm.hip recompiled
# under extra defines (A/B runs inside one gpurun call: MP3RGAIN_AMD_LIB=build_ab/libNAME.so python tools/...).
set -e
cd "$(dirname "$0")/../mp3rgain_amd/csrc"
name=$1; shift
mkdir -p ../../build_ab/obj_$name
COMMON="-O3 -Wno-missing-braces -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off "$@" -c rg_mp3dev.hip -o ../../build_ab/obj_$name/rg_mp3dev.o &
/opt/rocm/bin/hipcc $COMMON "$@" -c rg_mp3dev_host.hip -o ../../build_ab/obj_$name/rg_mp3dev_host.o &
/opt/rocm/bin/hipcc $COMMON "$@" -c rg_k2_tm.hip -o ../../build_ab/obj_$name/rg_k2_tm.o &
wait
objs=$(ls *.o | grep -v "^rg_mp3dev.o$" | grep -v "^rg_k2_tm.o$" | grep -v "^rg_mp3dev_host.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_ab/lib$name.so $objs ../../build_ab/obj_$name/rg_mp3dev.o ../../build_ab/obj_$name/rg_k2_tm.o ../../build_ab/obj_$name/rg_mp3dev_host.o -ldl
echo built build_ab/lib$name.so
