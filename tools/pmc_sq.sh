# Where the waves of the bench kernels spend their cycles: SQ counters in their own --pmc passes (no trace domains besides the
# kernel trace).  usage (on the GPU box): bash tools/pmc_sq.sh <fp32|bf16> <tag>
set -e
R=$GRAFT_REPO_ROOT
DT=$1
TAG=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
  -d $R/gpurun_out/pmc_sq1_$TAG -- python3 $R/bench.py --dtype $DT --groups 1 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > $R/gpurun_out/pmc_sq1_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM \
  -d $R/gpurun_out/pmc_sq2_$TAG -- python3 $R/bench.py --dtype $DT --groups 1 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > $R/gpurun_out/pmc_sq2_$TAG.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_FLAT \
  -d $R/gpurun_out/pmc_sq3_$TAG -- python3 $R/bench.py --dtype $DT --groups 1 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > $R/gpurun_out/pmc_sq3_$TAG.log 2>&1
python3 $R/tools/pmc_table.py $R/gpurun_out/pmc_sq1_$TAG $R/gpurun_out/pmc_sq2_$TAG $R/gpurun_out/pmc_sq3_$TAG > $R/gpurun_out/pmc_sq_$TAG.txt
cat $R/gpurun_out/pmc_sq_$TAG.txt
find $R/gpurun_out/pmc_sq?_$TAG -name "*.db" -size +20M -delete
