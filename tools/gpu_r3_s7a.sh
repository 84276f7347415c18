#!/bin/bash
# round 3, session 7a: wave-priority A/B for the main-stream kernels that share CUs with weight-gradient blocks
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s7a
run() { name=$1; shift; (env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>${O}_bench_$name.err | tail -1) > ${O}_bench_$name.json; python -c "import json;d=json.load(open('${O}_bench_$name.json'));print('$name',d['value'],d['ms_per_step'],d['config'].get('final_loss'))"; }
run base A=1
run norm SLAM_MAIN_PRIO=1
run attn SLAM_ATTN_PRIO=2
run both SLAM_MAIN_PRIO=1 SLAM_ATTN_PRIO=2
run base2 A=1
run both2 SLAM_MAIN_PRIO=1 SLAM_ATTN_PRIO=2
run norm2 SLAM_MAIN_PRIO=1
