cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p --output-format csv -- python $R/tools/stage_probe.py diffuse 8192 2 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p --output-format csv -- python $R/tools/stage_probe.py diffuse 8192 2 > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 -d $R/gpurun_out/pmc_sq -o p --output-format csv -- python $R/tools/stage_probe.py diffuse 8192 2 > $R/gpurun_out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_grbm -o p --output-format csv -- python $R/tools/stage_probe.py diffuse 8192 2 > $R/gpurun_out/pmc_grbm.log 2>&1
