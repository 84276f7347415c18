"""A/B of the staggered-phase kernels for the N = 896 launches (gemm_256x112 = 2: forced; edit the option key for
gemm_256x128) against the default 128 x 128 kernel, with the residual epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr(); dev = "cuda"; M = 8192
def timeit(fn, iters=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def rb(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
for name, (N, K) in {"o fwd": (896, 896), "qkv dgrad": (896, 1152), "down fwd": (896, 4864), "gate_up dgrad": (896, 9728)}.items():
    x, w, res = rb(M, K), rb(N, K), rb(M, N)
    ref = None
    for rep in range(2):
        for mode in (0, 2):
            lib.slam_set_option(None, b"gemm_256x112", mode)
            y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
            us = timeit(lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, res.data_ptr(), M, N, K, 2, st))
            if ref is None: ref = y
            print(f"{name:14s} 256x112={mode}: {us:8.1f} us {2.0*M*N*K/us/1e6:8.1f} TF same={torch.equal(ref, y)}", flush=True)
lib.slam_set_option(None, b"gemm_256x112", 0)
