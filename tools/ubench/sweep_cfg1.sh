cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --no-configs1 --no-mp3 --tracks-per-rank 1 --minutes 10 --steps 400 --warmup 20 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$TAG', 'ms/step %.4f'%d['ms_per_step'], 'frac %.4f'%r['frac'], 'kernel_ms %.4f'%r['kernel_ms'], 'conc %.2f'%r['kernel_concurrency'])
"; }
TAG="base L735"; run --tm-segment 735
TAG="nofix L735"; RG_DBG_NOFIX=1 run --tm-segment 735
TAG="8 streams, 8 queues L735"; GPU_MAX_HW_QUEUES=8 RG_DBG_STREAMS=8 run --tm-segment 735
TAG="8 streams, 8 queues L2205"; GPU_MAX_HW_QUEUES=8 RG_DBG_STREAMS=8 run --tm-segment 2205
TAG="8 streams, 8 queues auto"; GPU_MAX_HW_QUEUES=8 RG_DBG_STREAMS=8 run
TAG="8 streams, 4 queues L735"; RG_DBG_STREAMS=8 run --tm-segment 735
TAG="2 streams L735"; RG_DBG_STREAMS=2 run --tm-segment 735
TAG="2 streams L441"; RG_DBG_STREAMS=2 run --tm-segment 441
TAG="3 streams L735"; RG_DBG_STREAMS=3 run --tm-segment 735
TAG="base again"; run --tm-segment 735
