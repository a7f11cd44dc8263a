# Kernel statistics of the reference's own recipe (32x32 LR patches, batch 32, VGGStyleDiscriminator128) and of the C3 step; wall
# clock without the profiler first.  usage (GPU box): bash tools/prof_recipe.sh r03
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
O=$R/gpurun_out/prof_recipe_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=$R/tools/rocpd_summary.py
db() { find $1 -name "*_results.db" | head -1; }
for v in "fp32:--dtype fp32" "bf16:--dtype bf16 --disc-dtype bf16"; do
  k=${v%%:*}; fl=${v#*:}
  python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 10 --warmup 3 $fl > $O/wall_$k.json 2> $O/wall_$k.err
  rocprofv3 --kernel-trace --stats -d $O/ks_$k -- python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 4 --warmup 2 $fl > $O/ks_$k.log 2>&1
  python3 $S stats $(db $O/ks_$k) $O/${TAG}_recipe_${k}_kernel_stats.csv > $O/ks_$k.txt
  echo "== $k"; cat $O/wall_$k.json | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['value'])"
  head -12 $O/ks_$k.txt
done
find $O -name "*_results.db" -size +8M -delete
