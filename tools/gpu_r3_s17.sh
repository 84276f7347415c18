#!/bin/bash
# round 3, session 17: after the option clean-up - bit-identity of the two-stream backward, attention + norm kernels, bench
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/s17
(timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -k "side_stream or bucket or rmsnorm or attention_fwd_bwd or tiny" 2>&1 | tail -5) > ${O}_pytest.log; tail -3 ${O}_pytest.log
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>${O}_bench.err | tail -1) > ${O}_bench.json; python -c "import json;d=json.load(open('${O}_bench.json'));print(d['value'],d['ms_per_step'],d['config']['final_loss'])"
