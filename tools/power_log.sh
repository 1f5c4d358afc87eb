#!/bin/bash
# usage: tools/power_log.sh <outfile> <bench args...>   (GPU box)
# Samples rocm-smi (power, shader clock, temperature) every 0.2 s while a long bench run is in flight: the evidence
# for DESIGN.md's "FP64-FMA bound means power bound" (package power at the cap, shader clock below nominal).
OUTF=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
( while true; do echo "t=$(date +%s.%N) $(rocm-smi -d 0 --showpower --showclocks --showtemp --csv 2>/dev/null | tail -n +2 | tr '\n' ' ')"; sleep 0.2; done ) > $OUTF.raw &
SPID=$!
sleep 1
python bench.py --cpu-seconds 0 --no-configs1 "$@" > $OUTF.bench.json 2> $OUTF.err
sleep 1
kill $SPID
rocm-smi -d 0 --showpower --showclocks --showmaxpower --csv > $OUTF.header 2>/dev/null
cat $OUTF.header > $OUTF
cat $OUTF.raw >> $OUTF
rm -f $OUTF.raw $OUTF.header
wc -l $OUTF; tail -3 $OUTF; cat $OUTF.bench.json | head -c 600
