#!/bin/bash
# round 3, final session: the whole GPU suite, smoke, then the profiles the bench line is checked against (kernel trace +
# PMC passes on HEAD) and the bench lines of the three workloads
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/fin
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > ${O}_pytest.log; tail -4 ${O}_pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > ${O}_smoke.log; tail -2 ${O}_smoke.log
rm -rf gpurun_out/fin_prof gpurun_out/fin_prof_q15 gpurun_out/fin_pmc_*
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin_prof -o r3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/${O}_prof_bench.json 2> $R/${O}_prof_bench.err)
bash tools/pmc_step.sh fin slam358m
(timeout 500 python bench.py 2>${O}_bench_default.err | tail -1) > ${O}_bench_default.json
python -c "import json;d=json.load(open('${O}_bench_default.json'));print('default',d['value'],d['ms_per_step'],d['roofline']['frac'],d['cpu_baseline']['value'])"
(timeout 400 python bench.py --workload qwen1p5b --steps 6 --warmup 2 2>${O}_q15.err | tail -1) > ${O}_q15.json; python -c "import json;d=json.load(open('${O}_q15.json'));print('q15',d['value'],d['ms_per_step'])"
(timeout 300 python bench.py --workload dpo --steps 10 --warmup 3 2>${O}_dpo.err | tail -1) > ${O}_dpo.json; python -c "import json;d=json.load(open('${O}_dpo.json'));print('dpo',d['value'],d['ms_per_step'])"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin_prof_q15 -o q -- python $R/bench.py --workload qwen1p5b --steps 3 --warmup 1 > $R/${O}_prof_q15.json 2> $R/${O}_prof_q15.err)
