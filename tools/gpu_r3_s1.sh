#!/bin/bash
# round 3, session 1: new attention kernels - parity, same-box A/B against the round-2 library, kernel trace, step A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s1
(timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -25) > ${O}_pytest_attn.log; tail -3 ${O}_pytest_attn.log
(timeout 300 python tools/attn_bench.py --iters 30 --shapes 8x1024,1x8192,32x256 --libs r2,new --tunes 1.1.4,1.1.2,1.1.1,2.1.4,1.2.4,2.2.4,2.2.2 2>&1) > ${O}_attn_bench.log; cat ${O}_attn_bench.log
(timeout 200 python tools/attn_bench.py --iters 10 --hd 128 --heads 12,2 --shapes 4x2048 --libs r2,new 2>&1) > ${O}_attn_bench128.log; cat ${O}_attn_bench128.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s1_prof -o a -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 20 --shapes 8x1024 --libs r2,new --tunes 1.1.4,2.2.4 > $GRAFT_REPO_ROOT/${O}_prof.log 2>&1)
python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/s1_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
P
for lib in r2 new; do
  if [ $lib = r2 ]; then export SLAM_ENGINE_LIB=$PWD/slamkit_amd/lib/libslam_engine_r2.so; else unset SLAM_ENGINE_LIB; fi
  (timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>${O}_bench_$lib.err | tail -1) > ${O}_bench_$lib.json
  python -c "import json;d=json.load(open('${O}_bench_$lib.json'));print('$lib',d['value'],d['ms_per_step'],d['config'].get('final_loss'))"
done
unset SLAM_ENGINE_LIB
(timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "tiny or golden or wide or packed or padded or accumulates or likelihood or bit_identical" 2>&1 | tail -15) > ${O}_pytest_model.log; tail -3 ${O}_pytest_model.log
