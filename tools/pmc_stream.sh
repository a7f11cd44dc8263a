set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc_st1 -- python3 $R/tools/stream_bench.py > $R/gpurun_out/pmc_st1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_WAVES SQ_INSTS_MFMA -d $R/gpurun_out/pmc_st2 -- python3 $R/tools/stream_bench.py > $R/gpurun_out/pmc_st2.log 2>&1
python3 $R/tools/pmc_table.py $R/gpurun_out/pmc_st1 $R/gpurun_out/pmc_st2 > $R/gpurun_out/pmc_st.txt
find $R/gpurun_out/pmc_st? -name "*.db" -size +20M -delete
