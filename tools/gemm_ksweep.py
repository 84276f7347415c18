"""Fixed-overhead probe: NT GEMM time vs K at M=8192, N in {9728, 896} (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr(); M = 8192
def timeit(fn, iters=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for N in (9728, 896, 1152):
    for K in (64, 128, 256, 512, 896, 1792, 4864):
        x = (torch.randn(M, K, device="cuda") * .5).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * .5).to(torch.bfloat16)
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        us = timeit(lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 2, st))
        print(f"N={N:5d} K={K:5d} {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF  tiles={(M//128)*(N//128)}")
