#!/bin/bash
# round 3, session 19: last targeted check after removing the async image refresh
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/s19
(timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_train.py tests/test_gpu_model.py -x -q -k "dp or virtual_ranks or rccl or forced or side_stream or tiny or trajectory or fused_adamw" 2>&1 | tail -5) > ${O}_pytest.log; tail -3 ${O}_pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > ${O}_smoke.log; tail -1 ${O}_smoke.log
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>${O}_bench.err | tail -1) > ${O}_bench.json; python -c "import json;d=json.load(open('${O}_bench.json'));print(d['value'],d['ms_per_step'],d['config']['final_loss'],d['roofline']['frac'])"
STEPS=12 WARMUP=4 bash tools/dp1_bench.sh rsag:SLAM_DDP_ALGO=rs_ag
