# MFMA-busy and clock counters of the bench kernels (separate --pmc passes; no trace domains besides kernel-trace)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for dt in fp32 bf16; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_mfma_$dt -- python3 $R/bench.py --dtype $dt --groups 1 --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_mfma_$dt.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_gui_$dt -- python3 $R/bench.py --dtype $dt --groups 1 --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_gui_$dt.log 2>&1
  echo $dt done
done
