#!/bin/bash
# Counter passes over one program, each pass its own run: prof_pmc.sh <tag> <python script and arguments ...>
# Pass 1: SQ activity; pass 2: SQ memory/LDS; pass 3: FETCH_SIZE; pass 4: WRITE_SIZE.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
run() { # name, counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_${tag}_$name
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -o $name -- python3 "${PROG[@]}" > gpurun_out/pmc_${tag}_$name.log 2>&1
}
PROG=("$@")
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES && \
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM && \
run rd FETCH_SIZE && \
run wr WRITE_SIZE
