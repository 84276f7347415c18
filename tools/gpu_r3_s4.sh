#!/bin/bash
# round 3, session 4: full GPU suite on the instruction-diet attention + fused optimizer; kernel trace; A/Bs
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s4
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > ${O}_pytest.log; tail -6 ${O}_pytest.log
(timeout 200 python tools/attn_bench.py --iters 30 --shapes 8x1024,1x8192 --libs new --tunes 1.1.4,1.2.4,2.1.4 2>&1) > ${O}_attn_bench.log; cat ${O}_attn_bench.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s4_prof -o a -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 20 --shapes 8x1024 --libs new --tunes 1.1.4 > $GRAFT_REPO_ROOT/${O}_prof.log 2>&1)
python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/s4_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
P
run() { name=$1; shift; (env "$@" timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>${O}_bench_$name.err | tail -1) > ${O}_bench_$name.json; python -c "import json;d=json.load(open('${O}_bench_$name.json'));print('$name',d['value'],d['ms_per_step'],d['config'].get('final_loss'), [k['us'] for k in d['hbm_kernels']])"; }
run base A=1
run tile64 SLAM_ADAMW_TILE_COLS=64
run nofuse SLAM_FUSE_ADAMW_T=0
run base2 A=1
python -c "
import json;d=json.load(open('${O}_bench_base2.json'))
for k in d['roofline']['kernels']: print(k)
"
