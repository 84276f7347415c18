"""Square / long-K GEMM check of the 256 x 256 8-phase kernel against the 128 x 128 kernel (TFLOP/s on random data)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr(); dev = "cuda"
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in [(8192, 8192, 8192), (8192, 8192, 2048), (8192, 9728, 896), (16384, 17920, 1536)]:
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for rep in range(2):
        for g256 in (0, 2):
            lib.slam_set_option(None, b"gemm_256", g256)
            us = timeit(lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 2, st))
            print(f"M{M} N{N} K{K} g256={g256}: {us:9.1f} us {2.0*M*N*K/us/1e6:8.1f} TF", flush=True)
lib.slam_set_option(None, b"gemm_256", 1)
