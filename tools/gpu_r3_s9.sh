#!/bin/bash
# round 3, session 9: lean DMA-ring RMSNorm backward (<= 64 VGPRs: co-resident with weight-gradient GEMM blocks)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s9
(timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "rmsnorm" 2>&1 | tail -8) > ${O}_pytest.log; tail -4 ${O}_pytest.log
run() { name=$1; shift; (env "$@" timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>${O}_bench_$name.err | tail -1) > ${O}_bench_$name.json; python -c "import json;d=json.load(open('${O}_bench_$name.json'));print('$name',d['value'],d['ms_per_step'],d['config'].get('final_loss'), [(k['kernel'][:14],k['us']) for k in d['hbm_kernels']])"; }
P=$PWD/slamkit_amd/lib/libslam_engine_prev.so
run prev SLAM_ENGINE_LIB=$P
run new A=1
run prev2 SLAM_ENGINE_LIB=$P
run new2 A=1
runq() { name=$1; shift; (env "$@" timeout 300 python bench.py --workload qwen1p5b --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>${O}_q15_$name.err | tail -1) > ${O}_q15_$name.json; python -c "import json;d=json.load(open('${O}_q15_$name.json'));print('q15 $name',d['value'],d['ms_per_step'],d['config'].get('final_loss'))"; }
runq prev SLAM_ENGINE_LIB=$P
runq new A=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s9_prof -o r3 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/${O}_prof_bench.json 2> $GRAFT_REPO_ROOT/${O}_prof_bench.err)
(timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -x -q 2>&1 | tail -8) > ${O}_pytest2.log; tail -4 ${O}_pytest2.log
