"""The N = 896 NT launches of the Slam-358M step on the three kernels that can run them (stand-alone, HIP events):
128 x 128 (2 blocks per CU, one K-tile in flight per block), 256 x 224 (128 one-per-tile blocks: half the chip),
128 x 224 four-wave (256 blocks: one per CU).  python tools/probes/nt896_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from slamkit_amd import engine as E  # noqa: E402

lib = E.load_library()
st = E.current_stream_ptr()
dev = "cuda"
M = 8192
for name, N, K, use_res in (("down fwd + residual", 896, 4864, True), ("o fwd + residual", 896, 896, True), ("gate|up dgrad", 896, 9728, False),
                            ("qkv dgrad", 896, 1152, False), ("o dgrad", 896, 896, False)):
    X = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for kern, opts in (("128x128", {b"gemm_nt224": 0, b"gemm_nt128x224": 0}), ("256x224 (128 blocks)", {b"gemm_nt224": 2, b"gemm_nt128x224": 0}),
                       ("128x224 4-wave (256 blocks)", {b"gemm_nt224": 0, b"gemm_nt128x224": 2})):
        for k, v in opts.items():
            assert lib.slam_set_option(None, k, v) == 0
        f = lambda: lib.slam_op_gemm_nt(X.data_ptr(), W.data_ptr(), Y.data_ptr(), None, R.data_ptr() if use_res else None, M, N, K, 1, st)  # noqa: E731
        for _ in range(10):
            assert f() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40
        print(f"{name:22s} M{M} N{N} K{K:5d} {kern:30s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
lib.slam_set_option(None, b"gemm_nt224", 1)
lib.slam_set_option(None, b"gemm_nt128x224", 0)
