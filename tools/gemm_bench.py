"""GEMM micro-benchmark on the Slam-358M shapes (run on the GPU box): times the C-ABI op entry
points with HIP events. Usage: python tools/gemm_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E  # noqa: E402

lib = E.load_library()
M = 8192
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def rb(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


st = E.current_stream_ptr()
print(f"{'op':28s} {'mode':>5s} {'us':>9s} {'TFLOP/s':>9s}")
for name, (N, K) in {} if len(sys.argv) > 2 else {"qkv fwd": (1152, 896), "o fwd/dgrad": (896, 896), "gate_up fwd": (9728, 896),
                     "down fwd": (896, 4864), "qkv dgrad": (896, 1152), "down dgrad": (4864, 896),
                     "gate_up dgrad": (896, 9728)}.items():
    x, w, y = rb(M, K), rb(N, K), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for mode, gr in ((2, 4),):
        lib.slam_set_option(None, b"gemm_group_rows", gr)
        us = timeit(lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, mode, st))
        print(f"nt {name:25s} {mode:5d} gr={gr:2d} {us:9.1f} {2.0 * M * N * K / us / 1e6:9.1f}")
    lib.slam_set_option(None, b"gemm_group_rows", 4)
for name, (N, K) in {"wqkv": (1152, 896), "wo": (896, 896), "wgu": (9728, 896), "wd": (896, 4864)}.items():
    dy, x = rb(M, N), rb(M, K)
    ws = torch.empty(lib.slam_op_gemm_tn_workspace(M, N, K) // 4 + 16, dtype=torch.float32, device=dev)
    dw = torch.zeros(N, K, dtype=torch.float32, device=dev)
    ws = torch.empty(32 * N * K + 16, dtype=torch.float32, device=dev)
    for sk in (0, 1, 0, 1):
        lib.slam_set_option(None, b"gemm_tn_balanced", sk)
        us = timeit(lambda: lib.slam_op_gemm_tn(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), 1, M, N, K, ws.data_ptr(), st))
        print(f"tn {name:20s} balanced={sk} {us:9.1f} us {2.0 * M * N * K / us / 1e6:9.1f} TF (incl. reduce)")
    lib.slam_set_option(None, b"gemm_tn_balanced", 1)
    lib.slam_set_option(None, b"gemm_tn_splits", 0)
    lib.slam_set_option(None, b"gemm_tn_dma", 1)
