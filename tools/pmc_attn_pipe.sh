#!/bin/bash
# PMC passes over tools/pmc_attn.py for the pipelined attention kernels (default) and the one-chain kernels (WG_ATTN_PIPE=0); outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P="rocprofv3 --output-format csv --kernel-trace"
for v in 1 0; do
  export WG_ATTN_PIPE=$v
  $P --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d gpurun_out/pa_sq$v -o s -- python3 tools/pmc_attn.py > gpurun_out/pa_sq$v.log 2>&1
  $P --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d gpurun_out/pa_lds$v -o s -- python3 tools/pmc_attn.py > gpurun_out/pa_lds$v.log 2>&1
done
rocprofv3 -L > gpurun_out/counters.txt 2>&1
echo done
