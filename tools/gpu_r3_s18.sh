#!/bin/bash
# round 3, session 18: transposed weight images rebuilt under the next forward after a ranged optimizer step (rs_ag)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/s18
(timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_train.py -x -q -k "dp or virtual_ranks or rccl or rs_ag or forced" 2>&1 | tail -5) > ${O}_pytest.log; tail -3 ${O}_pytest.log
STEPS=12 WARMUP=4 bash tools/dp1_bench.sh rsag_async:SLAM_DDP_ALGO=rs_ag rsag_sync:SLAM_DDP_ALGO=rs_ag,SLAM_TREFRESH_ASYNC=0 rsag_async2:SLAM_DDP_ALGO=rs_ag rsag_sync2:SLAM_DDP_ALGO=rs_ag,SLAM_TREFRESH_ASYNC=0 allred:SLAM_DDP_ALGO=all_reduce
python -c "
import json
d=json.load(open('gpurun_out/dp1_rsag_async.json'));print(d['config'].get('exposed_param_gather_ms_3_steps'), d['config']['bucket_comm_ms'][:2])"
