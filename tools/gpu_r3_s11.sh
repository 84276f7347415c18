#!/bin/bash
# round 3, session 11: four-wave 256x256 kernel (128x128 per wave) in the step, A/B against the eight-wave persistent kernel
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s11
run() { name=$1; shift; (env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>${O}_bench_$name.err | tail -1) > ${O}_bench_$name.json; python -c "import json;d=json.load(open('${O}_bench_$name.json'));print('$name',d['value'],d['ms_per_step'],d['config'].get('final_loss'),d['roofline']['ms_per_launch'])"; }
run base A=1
run w4 SLAM_GEMM_256_W4=1
run base2 A=1
run w4b SLAM_GEMM_256_W4=1
