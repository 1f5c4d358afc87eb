#!/bin/bash
# [TARGET=units] tools/ab_chain.sh lib1 lib2 ... : tools/mp3_chain.py with each library in turn, twice (boards differ; compare
# within one call); TARGET = units per chunk (default 256 K; the file route's chunks are 768 K = 786432)
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    if [ "$lib" = default ]; then python tools/mp3_chain.py $TARGET 2>&1 | grep "ms per 256K"
    else MP3RGAIN_AMD_LIB=build_ab/lib$lib.so python tools/mp3_chain.py $TARGET 2>&1 | grep "ms per 256K"; fi
  done
done
