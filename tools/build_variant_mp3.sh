#!/bin/bash
# tools/build_variant_mp3.sh NAME "-DMACRO=.. ..."  -> build_ab/libNAME.so: the current library with only rg_mp3dev.hip recompiled
# under extra defines (A/B runs inside one gpurun call: MP3RGAIN_AMD_LIB=build_ab/libNAME.so python tools/...).
set -e
cd "$(dirname "$0")/../mp3rgain_amd/csrc"
name=$1; shift
mkdir -p ../../build_ab/obj_$name
/opt/rocm/bin/hipcc -O3 -Wno-missing-braces -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -c rg_mp3dev.hip -o ../../build_ab/obj_$name/rg_mp3dev.o
objs=$(ls *.o | grep -v "^rg_mp3dev.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_ab/lib$name.so $objs ../../build_ab/obj_$name/rg_mp3dev.o -ldl
echo built build_ab/lib$name.so
