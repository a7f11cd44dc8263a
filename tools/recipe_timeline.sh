set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04_tl
mkdir -p $O
db() { find $1 -name "*_results.db" | head -1; }
cd /tmp && export TMPDIR=/tmp
for d in fp32 bf16; do
  rocprofv3 --kernel-trace -d $O/kt_$d -- python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 6 --warmup 3 --dtype $d --disc-dtype $d > $O/kt_$d.log 2>&1
  echo "== recipe $d"
  python3 $R/tools/rocpd_summary.py timeline $(db $O/kt_$d) $O/tl_$d.txt | grep -v "^columns of\|^queue column"
done
find $R/gpurun_out -name "*_results.db" -size +8M -delete
