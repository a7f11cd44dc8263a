# The SQ counter passes of tools/pmc_sq.sh around an arbitrary python tool.  usage (on the GPU box): bash tools/pmc_sq_cmd.sh <tag> tools/<script>.py [args]
set -e
R=$GRAFT_REPO_ROOT
TAG=$1
shift
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
P3="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_FLAT"
rocprofv3 --pmc $P1 -d $R/gpurun_out/pmc_sq1_$TAG -- python3 $R/"$@" > $R/gpurun_out/pmc_sq1_$TAG.log 2>&1
rocprofv3 --pmc $P2 -d $R/gpurun_out/pmc_sq2_$TAG -- python3 $R/"$@" > $R/gpurun_out/pmc_sq2_$TAG.log 2>&1
rocprofv3 --pmc $P3 -d $R/gpurun_out/pmc_sq3_$TAG -- python3 $R/"$@" > $R/gpurun_out/pmc_sq3_$TAG.log 2>&1
python3 $R/tools/pmc_table.py $R/gpurun_out/pmc_sq1_$TAG $R/gpurun_out/pmc_sq2_$TAG $R/gpurun_out/pmc_sq3_$TAG > $R/gpurun_out/pmc_sq_$TAG.txt
cat $R/gpurun_out/pmc_sq_$TAG.txt
find $R/gpurun_out/pmc_sq?_$TAG -name "*.db" -size +20M -delete
