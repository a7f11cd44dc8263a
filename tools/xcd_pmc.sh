set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/xcd_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
db() { find $1 -name "*_results.db" | head -1; }
for m in 0 1; do
  export XCD_MAP=$m
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$m -- python3 $R/tools/wgrad_rdb_bench.py > $O/fetch_$m.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$m -- python3 $R/tools/wgrad_rdb_bench.py > $O/write_$m.log 2>&1
  python3 $R/tools/rocpd_summary.py traffic $O/traffic_$m.json $(db $O/fetch_$m) $(db $O/write_$m) > $O/traffic_$m.txt
  echo "mode $m"; cat $O/traffic_$m.json
done
