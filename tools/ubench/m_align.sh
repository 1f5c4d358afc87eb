for round in 1 2; do
for m in 0 32 16 30; do
  if [ $m = 0 ]; then A=""; else A="--tm-segment 2205 --tm-windows $m"; fi
  python bench.py --no-mp3 --cpu-seconds 0 --no-configs1 --no-one-shot $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('m=$m round $round: step %.3f ms frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
done; done
cd /tmp && export TMPDIR=/tmp
for m in 15 32; do
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc FETCH_SIZE -d /tmp/pm$m --output-format csv -- python bench.py --no-mp3 --cpu-seconds 0 --no-configs1 --no-one-shot --tm-segment 2205 --tm-windows $m --pre-roll 0.01 --steps 4 --warmup 1 > /dev/null 2>&1)
  python - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/pm$m/*/*counter_collection.csv") for r in csv.DictReader(open(f)) if "rg_tm_main" in r["Kernel_Name"]]
print("m=$m FETCH_SIZE mean KiB", sum(v)/len(v), "-> x2 GB", 2*sum(v)/len(v)*1024/1e9)
PY
done
