#!/bin/bash
# Shader clock and socket power WHILE the Slam-358M step runs (rocm-smi polled beside bench.py), against idle: is the step's
# throughput limited by the power / clock management of the part?  bash tools/probes/clock_under_load.sh [out.txt]
cd "${GRAFT_REPO_ROOT:-.}"; out=${1:-gpurun_out/clock_under_load.txt}
poll() { for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr -s ' \t' ' ' | tr '\n' '|'; echo; sleep 0.4; done; }
{
echo "== idle =="; poll 3
echo "== bench.py --steps 1500 (Slam-358M step, ~24 ms per step: polled from second 12 to second 24 of its ~45 s) =="
python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/clock_bench.json 2>/dev/null &
sleep 12; poll 24; wait
python -c "import json;d=json.load(open('gpurun_out/clock_bench.json'));print('bench', d['value'], d['ms_per_step'])"
echo "== 8192^3 GEMM loops (power_probe) =="
python tools/probes/power_probe.py 2>/dev/null | tail -16 &
sleep 7; poll 6; wait
} > $out 2>&1
cat $out
