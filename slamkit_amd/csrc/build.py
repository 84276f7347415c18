"""Build libslam_engine.so (gfx950) in-tree: hipcc per translation unit, then one shared link.

Usage: python -m slamkit_amd.csrc.build [--force] [--probes]

--probes builds lib/libslam_engine_probes.so with -DSLAM_PROBES instead: the ONLY build in which a GEMM can run without its
output stores (gemm_nt_store bit 1) or with another steady-state vmcnt (-DSLAM_PROBE_VMCNT via SLAM_PROBE_CFLAGS). Only
tools/probes/* load it (SLAM_ENGINE_LIB=...); the product library rejects those settings.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_DIR = os.path.join(ROOT, "lib")
LIB = os.path.join(LIB_DIR, "libslam_engine.so")
SOURCES = ["gemm.hip", "attention.hip", "elementwise.hip", "engine.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "slam_engine.h")]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950 has a unified file); without it the
# attention kernels spend 256 v_accvgpr_read/write per K/V tile moving the online-softmax state around.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


PROBES_LIB = os.path.join(LIB_DIR, "libslam_engine_probes.so")


def build(force=False, verbose=True, probes=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    objdir = os.path.join(HERE, "build_probes" if probes else "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    hipcc = _hipcc()
    FLAGS = list(globals()["FLAGS"]) + (["-DSLAM_PROBES"] + os.environ.get("SLAM_PROBE_CFLAGS", "").split() if probes else [])
    LIB = PROBES_LIB if probes else globals()["LIB"]

    def compile_one(src):
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, probes="--probes" in sys.argv))
