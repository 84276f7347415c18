#!/bin/bash
# PMC passes over one NT GEMM shape (run on the GPU box from the repo root): M N K as args
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/one_gemm.py <<PY
import os, sys, torch
sys.path.insert(0, "$R")
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr()
M, N, K = $1, $2, $3
x = (torch.randn(M, K, device="cuda") * .5).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * .5).to(torch.bfloat16)
y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(5): lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 2, st)
torch.cuda.synchronize()
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcg_$n -o p -- python /tmp/one_gemm.py > $R/gpurun_out/pmcg_$n.log 2>&1
done
