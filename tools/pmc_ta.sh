cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_BUSY_max" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES"; do
  n=$(echo $set | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmcta/$n --output-format csv -- python tools/mp3_chain.py 393216 > gpurun_out/pmcta_$n.log 2>&1 || echo "$n failed"
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcta/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].split('(')[0]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
    for k in acc:
        if 'backhalf' in k or 'huffman' in k:
            for c,v in acc[k].items():
                print(k[:28], c, sum(v)/len(v), len(v))
PY
