#!/bin/bash
# rocprofv3 kernel statistics of the data-parallel step on one GPU (1-rank RCCL group, every collective running).
R="${GRAFT_REPO_ROOT:-$PWD}"; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
export SLAM_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29731
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dp1prof -o dp1 -- \
  python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/dp1prof.log 2>&1
tail -2 $R/gpurun_out/dp1prof.log; find $R/gpurun_out/dp1prof -name "*kernel_stats.csv" | head
