# Round profile pass (on the GPU box): rocprofv3 kernel statistics of the bench commands and, in separate --pmc passes (no trace
# domains besides the kernel trace), HBM traffic, MFMA-busy and the SQ wave-state counters.  usage: bash tools/profile_round.sh r03
# bf16 traces run with --groups 1: per-kernel durations are only meaningful when launches do not overlap.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
STAGE=${2:-all}   # 1 = kernel statistics, 2 = PMC traffic / MFMA passes, 3 = SQ wave-state counters
O=$R/gpurun_out/prof_$TAG
mkdir -p $O $R/profiles
cd /tmp && export TMPDIR=/tmp
S=$R/tools/rocpd_summary.py
db() { find $1 -name "*_results.db" | head -1; }
run() { name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1 || { echo "FAILED $name"; tail -5 $O/$name.log; exit 1; }; echo "$name done"; }

# 1. kernel statistics
if [ "$STAGE" = all ] || [ "$STAGE" = 1 ]; then
run ks_default --kernel-trace --stats -d $O/ks_default -- python3 $R/bench.py --no-cpu-baseline
run ks_fp32 --kernel-trace --stats -d $O/ks_fp32 -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 10
run ks_bf16 --kernel-trace --stats -d $O/ks_bf16 -- python3 $R/bench.py --dtype bf16 --groups 1 --no-cpu-baseline --no-secondary --steps 10
export SR_BENCH_OVERLAP_G=0   # per-kernel durations are only meaningful when launches do not overlap: G's weight gradients on the caller's stream
run ks_c3 --kernel-trace --stats -d $O/ks_c3 -- python3 $R/bench.py --mode train --dtype bf16 --disc unet --lq 128 --batch 32 --steps 3 --warmup 1
unset SR_BENCH_OVERLAP_G
run ks_tiled --kernel-trace --stats -d $O/ks_tiled -- python3 $R/bench.py --mode tiled --dtype bf16 --steps 1 --warmup 1
run ks_tiled_fp32 --kernel-trace --stats -d $O/ks_tiled_fp32 -- python3 $R/bench.py --mode tiled --dtype fp32 --steps 1 --warmup 1
run ks_recipe_fp32 --kernel-trace --stats -d $O/ks_recipe_fp32 -- python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 4 --warmup 2 --dtype fp32
run ks_recipe_bf16 --kernel-trace --stats -d $O/ks_recipe_bf16 -- python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 4 --warmup 2 --dtype bf16 --disc-dtype bf16
python3 $S stats $(db $O/ks_default) $R/profiles/${TAG}_bench_default_kernel_stats.csv > $O/ks_default.txt
python3 $S stats $(db $O/ks_fp32) $R/profiles/${TAG}_fp32_kernel_stats.csv > $O/ks_fp32.txt
python3 $S stats $(db $O/ks_bf16) $R/profiles/${TAG}_bf16_kernel_stats.csv > $O/ks_bf16.txt
python3 $S stats $(db $O/ks_c3) $R/profiles/${TAG}_c3_train_bf16_unet_kernel_stats.csv > $O/ks_c3.txt
python3 $S stats $(db $O/ks_tiled) $R/profiles/${TAG}_c5_tiled_bf16_kernel_stats.csv > $O/ks_tiled.txt
python3 $S stats $(db $O/ks_tiled_fp32) $R/profiles/${TAG}_c5_tiled_fp32_kernel_stats.csv > $O/ks_tiled_fp32.txt
python3 $S stats $(db $O/ks_recipe_fp32) $R/profiles/${TAG}_recipe_fp32_kernel_stats.csv > $O/ks_recipe_fp32.txt
python3 $S stats $(db $O/ks_recipe_bf16) $R/profiles/${TAG}_recipe_bf16_kernel_stats.csv > $O/ks_recipe_bf16.txt
head -6 $O/ks_fp32.txt $O/ks_bf16.txt $O/ks_c3.txt
echo "stage 1 done"
fi

# 2. HBM traffic (FETCH_SIZE and WRITE_SIZE cannot share a pass) and MFMA busy
if [ "$STAGE" = all ] || [ "$STAGE" = 2 ]; then
B32="python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1"
B16="python3 $R/bench.py --dtype bf16 --groups 1 --no-cpu-baseline --no-secondary --steps 2 --warmup 1"
C3="python3 $R/bench.py --mode train --dtype bf16 --disc unet --lq 128 --batch 32 --steps 1 --warmup 1"
export SR_BENCH_OVERLAP_G=0   # counters per kernel: no second lane
TL="python3 $R/bench.py --mode tiled --dtype bf16 --steps 1 --warmup 0"
RF="python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 2 --warmup 1 --dtype fp32"
for w in f32:"$B32" b16:"$B16" c3:"$C3" tl:"$TL" rf:"$RF"; do
  k=${w%%:*}; cmd=${w#*:}
  run fetch_$k --pmc FETCH_SIZE -d $O/fetch_$k -- $cmd
  run write_$k --pmc WRITE_SIZE -d $O/write_$k -- $cmd
  run mfma_$k --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $O/mfma_$k -- $cmd
  run gui_$k --pmc GRBM_GUI_ACTIVE -d $O/gui_$k -- $cmd
done
# per-workload tables: the C3 step runs the same kernels at other sizes, so its dispatches must not be averaged into the inference rows
python3 $S traffic $R/profiles/traffic.json $(db $O/fetch_f32) $(db $O/write_f32) $(db $O/fetch_b16) $(db $O/write_b16) > $O/traffic.txt
python3 $S traffic $R/profiles/traffic_c3.json $(db $O/fetch_c3) $(db $O/write_c3) > $O/traffic_c3.txt
python3 $S traffic $R/profiles/traffic_tiled.json $(db $O/fetch_tl) $(db $O/write_tl) > $O/traffic_tiled.txt
python3 $S traffic $R/profiles/traffic_recipe_fp32.json $(db $O/fetch_rf) $(db $O/write_rf) > $O/traffic_recipe_fp32.txt
python3 $S mfma $R/profiles/mfma_util.json $(db $O/mfma_f32) $(db $O/gui_f32) $(db $O/mfma_b16) $(db $O/gui_b16) > $O/mfma.txt
python3 $S mfma $R/profiles/mfma_util_c3.json $(db $O/mfma_c3) $(db $O/gui_c3) > $O/mfma_c3.txt
python3 $S mfma $R/profiles/mfma_util_tiled.json $(db $O/mfma_tl) $(db $O/gui_tl) > $O/mfma_tiled.txt
python3 $S mfma $R/profiles/mfma_util_recipe_fp32.json $(db $O/mfma_rf) $(db $O/gui_rf) > $O/mfma_recipe_fp32.txt
for f in traffic traffic_c3 traffic_tiled traffic_recipe_fp32 mfma_util mfma_util_c3 mfma_util_tiled mfma_util_recipe_fp32; do cp $R/profiles/$f.json $R/profiles/${TAG}_$f.json; done

echo "stage 2 done"
fi
if [ "$STAGE" = all ] || [ "$STAGE" = 3 ]; then
# 3. wave-state counters of the inference kernels
bash $R/tools/pmc_sq.sh fp32 ${TAG}_fp32 > $O/sq_fp32.log 2>&1 && cp $R/gpurun_out/pmc_sq_${TAG}_fp32.txt $R/profiles/${TAG}_pmc_sq_fp32.txt
bash $R/tools/pmc_sq.sh bf16 ${TAG}_bf16 > $O/sq_bf16.log 2>&1 && cp $R/gpurun_out/pmc_sq_${TAG}_bf16.txt $R/profiles/${TAG}_pmc_sq_bf16.txt
fi
find $R/gpurun_out -name "*_results.db" -size +8M -delete
find $R/gpurun_out -name "*kernel_trace.csv" -delete
# gpurun brings back gpurun_out/ only: the summaries travel in a copy of profiles/ (copy it over profiles/ afterwards)
rm -rf $R/gpurun_out/profiles_out && mkdir -p $R/gpurun_out/profiles_out && cp $R/profiles/${TAG}_* $R/profiles/traffic*.json $R/profiles/mfma_util*.json $R/gpurun_out/profiles_out/
du -sh $R/gpurun_out $R/profiles
