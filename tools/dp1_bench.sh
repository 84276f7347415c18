#!/bin/bash
# The data-parallel step on ONE GPU with every collective of the N > 1 path running (1-rank RCCL group, SLAM_DP_FORCE=1):
# what the DP plumbing itself costs before any link is involved. Usage: bash tools/dp1_bench.sh name:ENV=VAL,... ...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}; [ "$envs" = "$cfg" ] && envs=""
  env SLAM_DP_FORCE=1 $(echo $envs | tr ',' ' ') timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $((29600 + RANDOM % 200)) bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-extras 2>gpurun_out/dp1_$name.err | tail -1 > gpurun_out/dp1_$name.json
  python - <<P
import json
try:
    d = json.load(open("gpurun_out/dp1_$name.json")); print("$name", d["value"], d["ms_per_step"], "exposed", d["config"]["exposed_comm_ms_last_step"])
except Exception as e:
    print("$name FAILED", e)
P
done
