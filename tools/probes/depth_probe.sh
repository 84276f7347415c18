#!/bin/bash
# prefetch-depth probe (see depth_probe.py); the probe libraries are built by hand:
#   hipcc <flags of csrc/build.py> -DSLAM_PROBE_VMCNT=2 -c gemm.hip -o gemm_p2.o; hipcc -shared ... -o lib/libslam_engine_probe2.so
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for l in libslam_engine.so libslam_engine_probe2.so libslam_engine_probe0.so libslam_engine.so; do
  SLAM_ENGINE_LIB=$PWD/slamkit_amd/lib/$l python tools/probes/depth_probe.py 2>&1 | grep TFLOP
done | tee gpurun_out/depth_probe.txt
