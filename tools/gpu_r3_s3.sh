#!/bin/bash
# round 3, session 3: parity of the DMA-diet attention, fused optimizer, rs_ag at one rank; A/Bs
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s3
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_gpu_dp.py -x -q -k "attention or adamw or optimizer or rs_ag or dp_forced or resume or trajectory" 2>&1 | tail -25) > ${O}_pytest.log; tail -4 ${O}_pytest.log
(timeout 200 python tools/attn_bench.py --iters 30 --shapes 8x1024,1x8192 --libs r2,new --tunes 1.1.4,1.2.4 2>&1) > ${O}_attn_bench.log; cat ${O}_attn_bench.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s3_prof -o a -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 20 --shapes 8x1024 --libs new --tunes 1.1.4 > $GRAFT_REPO_ROOT/${O}_prof.log 2>&1)
python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/s3_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
P
run() { name=$1; shift; (env "$@" timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>${O}_bench_$name.err | tail -1) > ${O}_bench_$name.json; python -c "import json;d=json.load(open('${O}_bench_$name.json'));print('$name',d['value'],d['ms_per_step'],d['config'].get('final_loss'))"; }
run base A=1
run nofuse SLAM_FUSE_ADAMW_T=0
run mixed SLAM_OPTIM_STATE_DTYPE=float32_bf16_moments
run bf16 SLAM_OPTIM_STATE_DTYPE=bfloat16
run base2 A=1
python -c "
import json;d=json.load(open('${O}_bench_base2.json'))
for k in d['roofline']['kernels']: print(k)
for k in d['hbm_kernels']: print(k['kernel'][:60], k['us'], k['GB_per_s'])
"
runq() { name=$1; shift; (env "$@" timeout 400 python bench.py --workload qwen1p5b --steps 6 --warmup 2 2>${O}_q_$name.err | tail -1) > ${O}_q_$name.json; python -c "import json;d=json.load(open('${O}_q_$name.json'));print('q15 $name',d['value'],d['ms_per_step'])"; }
runq base A=1
runq cus160 SLAM_BWD_WGRAD_CUS=160
runq cus192 SLAM_BWD_WGRAD_CUS=192
runq cus224 SLAM_BWD_WGRAD_CUS=224
