#!/bin/bash
# HBM traffic (TCC EA counters) of the dominant kernel: fused gate|up GEMM at the bench shape.
# Separate --pmc passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/one_gu.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr()
M, N, K = 8192, 9728, 896
x = (torch.randn(M, K, device="cuda") * .5).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * .02).to(torch.bfloat16)
y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); a = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
for _ in range(5): lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), a.data_ptr(), M, N, K, st)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmct_$c -o p -- python /tmp/one_gu.py > $R/gpurun_out/pmct_$c.log 2>&1
done
