#!/bin/bash
# One gpurun call: GPU parity suite + bench A/B over engine options + kernel trace. Usage (from the repo root on the GPU box):
#   bash tools/gpu_session.sh <tag> [configs...]     configs: name:ENV=VAL,ENV=VAL ...
# Results under gpurun_out/<tag>_*.
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${SKIP_PYTEST:-0}" != "1" ]; then
  (timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} 2>&1 | tail -60) > gpurun_out/${tag}_pytest.log
  tail -5 gpurun_out/${tag}_pytest.log
fi
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  [ "$envs" = "$cfg" ] && envs=""
  (env $(echo $envs | tr ',' ' ') timeout 400 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline ${BENCH_ARGS:-} 2>gpurun_out/${tag}_bench_${name}.err | tail -1) > gpurun_out/${tag}_bench_${name}.json
  python - <<P
import json
try:
    d = json.load(open("gpurun_out/${tag}_bench_${name}.json"))
    print("${name}", d["value"], d["ms_per_step"], d["config"].get("final_loss"))
except Exception as e:
    print("${name} FAILED", e)
P
done
