#!/bin/bash
# round 3, session 2: machine-model probes, PMC of the new attention kernels, priority / CU-mask step A/Bs
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/s2
(timeout 120 tools/probes/ubench 2>&1) > ${O}_ubench.log; cat ${O}_ubench.log
(timeout 200 python tools/attn_bench.py --iters 30 --shapes 8x1024 --libs new --tunes 1.1.4,1.1.4.1,1.2.4.1,1.1.3,1.1.3.1 2>&1) > ${O}_attn_bench.log; cat ${O}_attn_bench.log
R=$GRAFT_REPO_ROOT
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/s2_pmc_$n -o p -- python $R/tools/attn_bench.py --iters 3 --shapes 8x1024,1x8192 --libs new --tunes 1.1.4 > $R/gpurun_out/s2_pmc_$n.log 2>&1
done
cd $R
python - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/s2_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        name = k.split("(")[0].split("::")[-1] + " grid" + r["Grid_Size"]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c, v in sorted(acc[name].items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
P
run() { name=$1; shift; (env "$@" timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>${O}_bench_$name.err | tail -1) > ${O}_bench_$name.json; python -c "import json;d=json.load(open('${O}_bench_$name.json'));print('$name',d['value'],d['ms_per_step'],d['config'].get('final_loss'))"; }
run base A=1
run prio SLAM_ATTN_PRIO=1
run side SLAM_BENCH_STREAM=1
run cus96 SLAM_BWD_WGRAD_CUS=96
run cus128 SLAM_BWD_WGRAD_CUS=128
run cus160 SLAM_BWD_WGRAD_CUS=160
run cus192 SLAM_BWD_WGRAD_CUS=192
run base2 A=1
run prio_cus160 SLAM_ATTN_PRIO=1 SLAM_BWD_WGRAD_CUS=160
