# Counter passes over the extraction of 10 minutes of mono audio (each pass its own run).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_extract_sq -o sq -- python3 tools/extract_timing.py 26460000 > gpurun_out/pmc_extract_sq.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_extract_sq2 -o sq2 -- python3 tools/extract_timing.py 26460000 > gpurun_out/pmc_extract_sq2.log 2>&1
