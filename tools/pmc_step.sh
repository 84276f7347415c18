#!/bin/bash
# PMC passes over a whole optimizer step (tools/one_step.py: every kernel of the hot path in one go), kernel-trace only, one
# counter set per pass (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; SQ has 8 slots). Run on the GPU box:
#   bash tools/pmc_step.sh <tag> [slam358m|qwen1p5b]   -> gpurun_out/<tag>_pmc_<SET>/ ; summarise with tools/pmc_summary.py
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/${tag}_pmc_$i -o p -- \
    python $R/tools/one_step.py "$@" > $R/gpurun_out/${tag}_pmc_$i.log 2>&1
  ls $R/gpurun_out/${tag}_pmc_$i/*/ 2>/dev/null | head -3
done
