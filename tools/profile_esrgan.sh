set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b_esrgan_bf16 -- python3 $R/tools/perf_esrgan_step.py training_config/train_rrdbnet_esrgan_x4_mi355x.yml 32 32 5 bf16 > $R/gpurun_out/r01b_esrgan_bf16.log 2>&1
tail -2 $R/gpurun_out/r01b_esrgan_bf16.log
find $R/gpurun_out -name "*kernel_trace.csv" -delete
