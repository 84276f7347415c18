#!/bin/bash
# round 3, session 14: sharded-optimizer ranges over virtual ranks, both exchange algorithms on a 1-rank RCCL group through
# bench.py's own multi-rank code path (per-bucket collective times), DP tests
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s14
(timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "virtual_ranks or fused_adamw" 2>&1 | tail -6) > ${O}_pytest.log; tail -3 ${O}_pytest.log
(timeout 900 python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -6) > ${O}_pytest_dp.log; tail -3 ${O}_pytest_dp.log
STEPS=12 WARMUP=4 bash tools/dp1_bench.sh rsag:SLAM_DDP_ALGO=rs_ag allred:SLAM_DDP_ALGO=all_reduce rsag32:SLAM_DDP_ALGO=rs_ag,SLAM_DDP_COMM_DTYPE=float32
python - <<'P'
import json
for n in ("rsag", "allred", "rsag32"):
    try:
        d = json.load(open(f"gpurun_out/dp1_{n}.json")); c = d["config"]
        print(n, c["ddp_algo"], "exposed", c["exposed_comm_ms_last_step"], "gather", c["exposed_param_gather_ms_3_steps"], c["bucket_comm_ms"])
    except Exception as e:
        print(n, "FAILED", e)
P
(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>${O}_plain.err | tail -1) > ${O}_plain.json; python -c "import json;d=json.load(open('${O}_plain.json'));print('plain',d['value'],d['ms_per_step'])"
