#!/bin/bash
# tools/ab_bench.sh lib1 lib2 ... : bench.py (headline, one-shot, configs[1]) with each library in turn, twice -- boards of the
# pool differ by several per cent, so builds are compared inside ONE gpurun call.  "default" = the in-tree library.
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset MP3RGAIN_AMD_LIB; else export MP3RGAIN_AMD_LIB=build_ab/lib$lib.so; fi
    python bench.py --no-mp3 --cpu-seconds 0 ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d.get('one_shot') or {}; c=d.get('configs1') or {}
print('== %-10s round $round: step %.3f ms frac %.4f | one-shot call %.3f ms kernel %.3f | configs1 %.2f us frac %.4f' % ('$lib', d['ms_per_step'], d['roofline']['frac'], o.get('ms_per_call',0), o.get('kernel_ms_alone',0), c.get('ms_per_step',0)*1e3, (c.get('roofline') or {}).get('frac',0)))"
  done
done
