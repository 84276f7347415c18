#!/bin/bash
# End-of-round measurement session: full GPU suite, smoke, the driver's bench command, kernel trace + PMC passes of the step,
# the configs[3]-shaped workload. Results under gpurun_out/<tag>_*; summaries are made from them on the authoring side
# (tools/prof_summary.py, tools/trace_breakdown.py, tools/pmc_summary.py) and committed under profiles/.
tag=${1:-fin}
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/${tag}
[ "${SKIP_FULL:-0}" = 1 ] || bash tools/gpu_run.sh ${tag} full
bash tools/gpu_run.sh ${tag} smoke
(timeout 600 python bench.py 2>${O}_bench_default.err | tail -1) > ${O}_bench_default.json
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>${O}_bench_driver.err | tail -1) > ${O}_bench_driver.json   # the driver's round-end command
python -c "import json;d=json.load(open('${O}_bench_driver.json'));print('bench (driver command)', d['value'], d['ms_per_step'], d['config']['ms_per_step_median_50'])"
python -c "import json;d=json.load(open('${O}_bench_default.json'));print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])"
cd /tmp
SLAM_BENCH_MEDIAN_STEPS=5 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_prof -o r6 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > ${O}_prof.log 2>&1
cd $R; find ${O}_prof -name "*.csv" | head
bash tools/pmc_step.sh ${tag} slam358m
(timeout 400 python bench.py --workload qwen1p5b --steps 5 --warmup 2 2>${O}_bench_q.err | tail -1) > ${O}_bench_q.json
python -c "import json;d=json.load(open('${O}_bench_q.json'));print('qwen', d['value'], d['ms_per_step'])"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_profq -o r6q -- python $R/bench.py --workload qwen1p5b --steps 3 --warmup 1 > ${O}_profq.log 2>&1
cd $R; find ${O}_profq -name "*.csv" | head -4
(timeout 300 python bench.py --workload dpo --steps 10 --warmup 3 2>${O}_bench_dpo.err | tail -1) > ${O}_bench_dpo.json
python -c "import json;d=json.load(open('${O}_bench_dpo.json'));print('dpo', d['value'], d['ms_per_step'])"
# the data-parallel path with every collective on a 1-rank RCCL group (exchange-variant table included), and the 2-rank gloo plumbing check
STEPS=20 WARMUP=5 bash tools/dp1_bench.sh rsag_bf16:SLAM_DDP_ALGO=rs_ag
python -c "import json;d=json.load(open('gpurun_out/dp1_rsag_bf16.json'));print('dp1 variants', json.dumps(d.get('extras', {}).get('dp_variants'))[:1200])"
bash tools/gpu_run.sh ${tag} dp2
