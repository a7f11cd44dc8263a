# Development aid: the bf16 inference bench with the inference forward's ring-only stores of x1..x4 on (0) and forced off (-1):
# kernel durations and, in their own passes, the HBM bytes written / fetched per launch of the fused dense block.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in -1 0; do
  export SR_DEV_MIDS_SCRATCH=$v
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $c -d $R/gpurun_out/ab_pmc_${c}_$v -- python3 $R/bench.py --dtype bf16 --groups 1 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > $R/gpurun_out/ab_pmc_${c}_$v.log 2>&1
  done
  python3 $R/tools/rocpd_summary.py traffic /tmp/traffic_$v.json $(find $R/gpurun_out/ab_pmc_FETCH_SIZE_$v -name "*_results.db" | head -1) $(find $R/gpurun_out/ab_pmc_WRITE_SIZE_$v -name "*_results.db" | head -1) > /tmp/traffic_$v.txt
  echo "SR_DEV_MIDS_SCRATCH=$v"; grep -i "rdb_fused" /tmp/traffic_$v.txt | head -3; python3 -c "
import json; d=json.load(open('/tmp/traffic_$v.json'))
for k,v in d.items():
    if 'rdb_fused' in k: print(k[:60], v)
"
done
find $R/gpurun_out -name "*_results.db" -delete
