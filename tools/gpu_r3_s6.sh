#!/bin/bash
# round 3, session 6: pre-scaled-query attention (single rounding) - full GPU suite, kernel trace, step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s6
(timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -15) > ${O}_pytest_attn.log; tail -3 ${O}_pytest_attn.log
(timeout 1700 python -m pytest tests -m gpu -x -q -k "not attention" 2>&1 | tail -30) > ${O}_pytest.log; tail -6 ${O}_pytest.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s6_prof -o a -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 20 --shapes 8x1024 --libs new --tunes 1.1.4,2.1.4 > $GRAFT_REPO_ROOT/${O}_prof.log 2>&1)
python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/s6_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print(r["Name"][:78], r["Calls"], r["AverageNs"])
P
(timeout 200 python tools/attn_bench.py --iters 30 --shapes 8x1024,1x8192 --libs new --tunes 1.1.4,2.1.4 2>&1) > ${O}_attn_bench.log; cat ${O}_attn_bench.log
run() { name=$1; shift; (env "$@" timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>${O}_bench_$name.err | tail -1) > ${O}_bench_$name.json; python -c "import json;d=json.load(open('${O}_bench_$name.json'));print('$name',d['value'],d['ms_per_step'],d['config'].get('final_loss'), [k['us'] for k in d['hbm_kernels']])"; }
run base A=1
run jq2 SLAM_ATTN_JQ=2
run base2 A=1
