#!/bin/bash
# round 3, last check on HEAD: the whole GPU suite + smoke + one default bench line
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/fin2
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > ${O}_pytest.log; tail -4 ${O}_pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > ${O}_smoke.log; tail -2 ${O}_smoke.log
(timeout 500 python bench.py 2>${O}_bench_default.err | tail -1) > ${O}_bench_default.json
python -c "import json;d=json.load(open('${O}_bench_default.json'));print('default',d['value'],d['ms_per_step'],d['roofline']['frac'],d['cpu_baseline']['value'])"
