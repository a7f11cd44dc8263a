set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "icache\|ifetch\|INST_CACHE\|SQC_" | head -30 > $R/gpurun_out/icache_counters.txt || true
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  rocprofv3 --pmc $c -d $R/gpurun_out/icache_$(echo $c | cut -d' ' -f1) -- python3 $R/bench.py --dtype bf16 --groups 1 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > $R/gpurun_out/icache.log 2>&1 || echo "failed $c"
done
python3 - <<'PY'
import sqlite3,glob,os
R=os.environ['GRAFT_REPO_ROOT']
for db in glob.glob(R+'/gpurun_out/icache_*/**/*_results.db', recursive=True):
    con=sqlite3.connect(db)
    tabs=[r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    pmc=[t for t in tabs if 'pmc_event' in t][0]; info=[t for t in tabs if 'info_pmc' in t][0]; disp=[t for t in tabs if 'kernel_dispatch' in t][0]; sym=[t for t in tabs if 'info_kernel_symbol' in t][0]
    q=f"select s.kernel_name, i.name, sum(p.value), count(distinct d.id) from {pmc} p join {info} i on p.pmc_id=i.id join {disp} d on p.event_id=d.event_id join {sym} s on d.kernel_id=s.id where s.kernel_name like '%rdb_fused%' group by 1,2"
    try:
        for r in con.execute(q): print(r[0][:50], r[1], r[2]/max(r[3],1), 'per dispatch over', r[3])
    except Exception as e: print('query failed', e, tabs[:8])
PY
find $R/gpurun_out -name "*_results.db" -delete
