#!/bin/bash
# One gpurun call (rounds 4 - 6). Usage on the GPU box, from the repo root:
#   bash tools/gpu_run.sh <tag> <stage> [<stage> ...]
# stages:  t:<pytest -k expression>   targeted GPU tests        full          the whole -m gpu suite
#          b:<name>[:ENV=V,ENV=V]     bench.py A/B line         smoke         __graft_entry__.smoke()
#          q:<name>[:ENV=V,...]       bench.py --workload qwen1p5b
#          p:<script.py>[:args]       a tools/probes/ probe     prof / pmc    kernel trace / counters of bench.py
#          x:<script.py>[:args]       a tools/ script           pos           tools/probes/power_or_stall.py under one PMC pass
# Every stage runs under its own timeout; results under gpurun_out/<tag>_*.
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/${tag}
for st in "$@"; do
  kind=${st%%:*}; rest=${st#*:}; [ "$rest" = "$st" ] && rest=""
  case $kind in
    t) L=${O}_t_$(echo "$rest" | tr -c 'a-zA-Z0-9' '_' | cut -c1-40).log
       (timeout ${T_TIMEOUT:-900} python -m pytest tests -m gpu -x -q -s -k "$rest" 2>&1 | grep -E "parity|passed|failed|Error|error|assert" | tail -${T_TAIL:-40}) > $L; tail -n 3 $L;;
    full) (timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -60) > ${O}_pytest.log; tail -5 ${O}_pytest.log;;
    smoke) (timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > ${O}_smoke.log; tail -1 ${O}_smoke.log;;
    b|q) name=${rest%%:*}; envs=${rest#*:}; [ "$envs" = "$rest" ] && envs=""
       wl=""; [ "$kind" = q ] && wl="--workload qwen1p5b"
       (env $(echo $envs | tr ',' ' ') timeout 500 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline ${BENCH_ARGS:---no-extras} $wl 2>${O}_bench_${name}.err | tail -1) > ${O}_bench_${name}.json
       python - <<P
import json
try:
    d = json.load(open("${O}_bench_${name}.json"))
    r = d.get("roofline", {})
    print("${name}", d["value"], d["ms_per_step"], d["config"].get("final_loss"), r.get("frac"), r.get("dominant_by_time"))
except Exception as e:
    print("${name} FAILED", e)
P
       ;;
    p) scr=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
       (timeout 600 python tools/probes/$scr $(echo $args | tr ',' ' ') 2>&1 | tail -80) > ${O}_probe_$(basename $scr .py).log; tail -3 ${O}_probe_$(basename $scr .py).log;;
    x) scr=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
       (timeout ${X_TIMEOUT:-600} python tools/$scr $(echo $args | tr ',' ' ') 2>&1 | tail -${X_TAIL:-80}) > ${O}_x_$(basename $scr .py).log; tail -3 ${O}_x_$(basename $scr .py).log;;
    pos) R=$(pwd); cd /tmp
       timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA \
         --output-format csv -d $R/${O}_pos -o p -- python $R/tools/probes/power_or_stall.py $R/${O}_pos_manifest.json > $R/${O}_pos.log 2>&1
       cd $R; python tools/probes/power_or_stall_summary.py ${O}_pos ${O}_pos_manifest.json ${O}_pos.md | tail -14;;
    pmcx) R=$(pwd); scr=${rest%%:*}; cd /tmp   # pmcx:<tools/probes script writing a manifest>: one PMC pass + the power_or_stall summary
       timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA \
         --output-format csv -d $R/${O}_pmcx -o p -- python $R/tools/probes/$scr $R/${O}_pmcx_manifest.json > $R/${O}_pmcx.log 2>&1
       cd $R; python tools/probes/power_or_stall_summary.py ${O}_pmcx ${O}_pmcx_manifest.json ${O}_pmcx.md | tail -14;;
    dp2) # plumbing check of bench.py's N > 1 path on the one GPU: two ranks on device 0, exchange over gloo (NOT a measurement)
       (env SLAM_BENCH_BACKEND=gloo SLAM_BENCH_DEVICE=0 SLAM_BENCH_MEDIAN_STEPS=3 SLAM_BENCH_DP_VARIANT_STEPS=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
          --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>${O}_dp2.err | tail -1) > ${O}_dp2.json
       python -c "import json;d=json.load(open('${O}_dp2.json'));print('dp2', d['n_gpus'], d['value'], d['config'].get('ms_per_step_median_50'), json.dumps(d.get('extras'))[:1500])" || tail -5 ${O}_dp2.err;;
    prof) cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/${O}_prof -o r4 -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OLDPWD/${O}_prof.log 2>&1; cd $OLDPWD; ls ${O}_prof | head;;
    *) echo "unknown stage $st";;
  esac
done
