for m in 12 15 16 18 20 24 25 30 36 40 45 48 50 60 72 75 90; do
  python bench.py --no-mp3 --cpu-seconds 0 --no-configs1 --no-one-shot --tm-segment 2205 --tm-windows $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('m=%3d stride %8d B (mod 32768 = %5d): step %.3f ms frac %.4f' % ($m, $m*8820, ($m*8820)%32768, d['ms_per_step'], d['roofline']['frac']))"
done
