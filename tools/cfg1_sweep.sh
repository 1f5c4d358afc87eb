for args in "" "--tm-segment 441" "--tm-segment 735" "--tm-segment 2205" "--slots 4" "--slots 6" "--slots 12" "--tm-segment 2205 --slots 12"; do
  echo "== $args"
  python bench.py --tracks-per-rank 1 --minutes 10 --steps 400 --warmup 20 --no-configs1 --no-mp3 --cpu-seconds 0 $args 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(o['ms_per_step']*1000,2),'us', round(o['roofline']['frac'],4))"
done
