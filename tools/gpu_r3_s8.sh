#!/bin/bash
# round 3, session 8: backward with no side->main edges (gradients in dead activation buffers), batched main->side
# hand-overs, events without the system-scope fence: parity tests, then a same-box A/B against the previous library
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s8
(timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_dp.py -x -q 2>&1 | tail -8) > ${O}_pytest.log; tail -4 ${O}_pytest.log
(timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "side_stream or bucket or clip_and_adamw" 2>&1 | tail -8) > ${O}_pytest2.log; tail -4 ${O}_pytest2.log
run() { name=$1; shift; (env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>${O}_bench_$name.err | tail -1) > ${O}_bench_$name.json; python -c "import json;d=json.load(open('${O}_bench_$name.json'));print('$name',d['value'],d['ms_per_step'],d['config'].get('final_loss'))"; }
P=$PWD/slamkit_amd/lib/libslam_engine_prev.so
run prev SLAM_ENGINE_LIB=$P
run f4 SLAM_BWD_WGRAD_FORKS=4
run f4fence SLAM_BWD_WGRAD_FORKS=4 SLAM_EVENT_SYSTEM_FENCE=1
run f2 SLAM_BWD_WGRAD_FORKS=2
run f1 SLAM_BWD_WGRAD_FORKS=1
run prev2 SLAM_ENGINE_LIB=$P
run f4b SLAM_BWD_WGRAD_FORKS=4
run f2b SLAM_BWD_WGRAD_FORKS=2
run f1b SLAM_BWD_WGRAD_FORKS=1
run f2fence SLAM_BWD_WGRAD_FORKS=2 SLAM_EVENT_SYSTEM_FENCE=1
(env SLAM_BWD_WGRAD_FORKS=2 timeout 300 python bench.py --workload qwen1p5b --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>${O}_q15.err | tail -1) > ${O}_q15.json; python -c "import json;d=json.load(open('${O}_q15.json'));print('q15 f2',d['value'],d['ms_per_step'])"
(env SLAM_ENGINE_LIB=$P timeout 300 python bench.py --workload qwen1p5b --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>${O}_q15p.err | tail -1) > ${O}_q15p.json; python -c "import json;d=json.load(open('${O}_q15p.json'));print('q15 prev',d['value'],d['ms_per_step'])"
