# bf16 traces run with --groups 1: per-kernel durations are only meaningful when launches do not overlap (the default
# bf16 forward runs image groups on concurrent streams)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b_fp32 -- python3 $R/bench.py --no-cpu-baseline --steps 10 > $R/gpurun_out/r01b_fp32.log 2>&1
echo fp32 trace done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b_bf16 -- python3 $R/bench.py --dtype bf16 --groups 1 --no-cpu-baseline --steps 10 > $R/gpurun_out/r01b_bf16.log 2>&1
echo bf16 trace done
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r01b_pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r01b_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r01b_pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r01b_pmc_write.log 2>&1
echo fp32 pmc done
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r01b_pmc_fetch16 -- python3 $R/bench.py --dtype bf16 --groups 1 --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r01b_pmc_fetch16.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r01b_pmc_write16 -- python3 $R/bench.py --dtype bf16 --groups 1 --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r01b_pmc_write16.log 2>&1
echo bf16 pmc done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b_train_bf16 -- python3 $R/tools/perf_train.py 16 bf16 > $R/gpurun_out/r01b_train_bf16.log 2>&1
echo train bf16 done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b_train_fp32 -- python3 $R/tools/perf_train.py 16 fp32 > $R/gpurun_out/r01b_train_fp32.log 2>&1
echo train fp32 done
find $R/gpurun_out -name "*kernel_trace.csv" -delete
du -sh $R/gpurun_out
