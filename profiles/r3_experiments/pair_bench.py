"""Down-projection forward shape (M 8192, N 896, K 4864, residual epilogue) on the three NT paths: 128x128 kernel, 256x224
kernel one block per tile (128 blocks), 256x224 K-pair launch (256 blocks). HIP-event timing, standalone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slamkit_amd import engine as E

lib = E.load_library()
st = E.current_stream_ptr()
dev = "cuda"
shapes = [(8192, 896, 4864), (8192, 896, 9728), (8192, 896, 1152)]
for M, N, K in shapes:
    X = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for name, opts in (("128x128", {b"gemm_nt224": 0, b"gemm_nt224_pair": 0}), ("256x224 x128 blocks", {b"gemm_nt224": 2, b"gemm_nt224_pair": 0}),
                       ("256x224 K-pair", {b"gemm_nt224": 1, b"gemm_nt224_pair": 2})):
        for k, v in opts.items():
            assert lib.slam_set_option(None, k, v) == 0
        f = lambda: lib.slam_op_gemm_nt(X.data_ptr(), W.data_ptr(), Y.data_ptr(), None, R.data_ptr(), M, N, K, 1, st)
        for _ in range(10):
            assert f() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"M{M} N{N} K{K} {name:22s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
    lib.slam_set_option(None, b"gemm_nt224", 1); lib.slam_set_option(None, b"gemm_nt224_pair", 1)
