# Kernel statistics + host issue time of the reference's own recipe (train_ESRGAN_x4.yml shapes), fp32 and bf16.
# usage (GPU box): bash tools/recipe_profile.sh r04
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
S=$R/tools/rocpd_summary.py
db() { find $1 -name "*_results.db" | head -1; }
python3 $R/tools/issue_time.py training_config/train_rrdbnet_esrgan_x4_mi355x.yml 32 32 20 fp32 fp32 > $O/issue_fp32.txt 2>&1
python3 $R/tools/issue_time.py training_config/train_rrdbnet_esrgan_x4_mi355x.yml 32 32 20 bf16 bf16 > $O/issue_bf16.txt 2>&1
cat $O/issue_fp32.txt $O/issue_bf16.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ks_recipe_fp32 -- python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 4 --warmup 2 --dtype fp32 > $O/ks_recipe_fp32.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/ks_recipe_bf16 -- python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 4 --warmup 2 --dtype bf16 --disc-dtype bf16 > $O/ks_recipe_bf16.log 2>&1
python3 $S stats $(db $O/ks_recipe_fp32) $R/gpurun_out/${TAG}_recipe_fp32_kernel_stats.csv > $O/ks_recipe_fp32.txt
python3 $S stats $(db $O/ks_recipe_bf16) $R/gpurun_out/${TAG}_recipe_bf16_kernel_stats.csv > $O/ks_recipe_bf16.txt
head -30 $O/ks_recipe_fp32.txt $O/ks_recipe_bf16.txt
find $R/gpurun_out -name "*_results.db" -size +8M -delete
find $R/gpurun_out -name "*kernel_trace.csv" -delete
