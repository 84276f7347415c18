"""A/B of the first-round stagger of the 256 x 256 kernel on its two launches of the Slam-358M layer (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr(); dev = "cuda"
M = 8192
def timeit(fn, iters=30):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def rb(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
x, w = rb(M, 896), rb(9728, 896) * 0.04
y, act = torch.empty(M, 9728, dtype=torch.bfloat16, device=dev), torch.empty(M, 4864, dtype=torch.bfloat16, device=dev)
for groups in (2, 3, 4):
    for stg in (0, 1, 2, 3, 4, 6):
        if stg == 0 and groups != 2: continue
        lib.slam_set_option(None, b"gemm_256_stagger", stg); lib.slam_set_option(None, b"gemm_256_stagger_groups", groups)
        us = timeit(lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, 9728, 896, st))
        usp = timeit(lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, 9728, 896, 1, st))
        print(f"gate|up fwd groups {groups} stagger {stg}: fused {us:7.1f} us {2.0*M*9728*896/us/1e6:7.1f} TF   plain {usp:7.1f} us", flush=True)
lib.slam_set_option(None, b"gemm_256_stagger", 0)
