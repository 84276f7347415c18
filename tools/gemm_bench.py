"""GEMM micro-benchmark on the Slam-358M shapes (run on the GPU box): times the C-ABI op entry points with HIP events,
interleaved A/B of the wgrad kernels (gemm_tn224 0/1). Usage: python tools/gemm_bench.py [iters] [qwen]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E  # noqa: E402

lib = E.load_library()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
qwen = len(sys.argv) > 2
M = 16384 if qwen else 8192
H, I, QKV = (1536, 8960, 2048) if qwen else (896, 4864, 1152)
dev = "cuda"


def timeit(fn):
    for _ in range(5):
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
print(f"{'op':28s} {'us':>9s} {'TFLOP/s':>9s}")
for name, (N, K) in {"qkv fwd": (QKV, H), "o fwd/dgrad": (H, H), "gate_up fwd": (2 * I, H), "down fwd": (H, I),
                     "qkv dgrad": (H, QKV), "down dgrad": (I, H), "gate_up dgrad": (H, 2 * I)}.items():
    x, w, y = rb(M, K), rb(N, K), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 1, st))
    print(f"nt {name:25s} {us:9.1f} {2.0 * M * N * K / us / 1e6:9.1f}")
for name, (N, K) in {"wqkv": (QKV, H), "wo": (H, H), "wgu": (2 * I, H), "wd": (H, I)}.items():
    dy, x = rb(M, N), rb(M, K)
    lib.slam_set_option(None, b"gemm_tn224", 2)
    ws = torch.empty(lib.slam_op_gemm_tn_workspace(M, N, K) // 4 + 16, dtype=torch.float32, device=dev)
    dw = torch.zeros(N, K, dtype=torch.float32, device=dev)
    for mode in (0, 1, 0, 1):
        lib.slam_set_option(None, b"gemm_tn224", mode)
        for acc in (0, 1):
            us = timeit(lambda: lib.slam_op_gemm_tn(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), acc, M, N, K, ws.data_ptr(), st))
            print(f"tn {name:8s} tn224={mode} acc={acc} {us:9.1f} us {2.0 * M * N * K / us / 1e6:9.1f} TF (incl. reduce)")
    lib.slam_set_option(None, b"gemm_tn224", 1)
if os.environ.get("LONGM"):
    # main-loop rates without the per-piece prologue / epilogue: one long contraction
    Ml = 65536
    for name, (N, K) in {"wgu long-M": (2 * I, H), "wd long-M": (H, I)}.items():
        dy, x = rb(Ml, N), rb(Ml, K)
        lib.slam_set_option(None, b"gemm_tn224", 2)
        ws = torch.empty(lib.slam_op_gemm_tn_workspace(Ml, N, K) // 4 + 16, dtype=torch.float32, device=dev)
        dw = torch.zeros(N, K, dtype=torch.float32, device=dev)
        for mode in (0, 1, 0, 1):
            lib.slam_set_option(None, b"gemm_tn224", mode)
            us = timeit(lambda: lib.slam_op_gemm_tn(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), 0, Ml, N, K, ws.data_ptr(), st))
            print(f"tn {name:12s} tn224={mode} {us:9.1f} us {2.0 * Ml * N * K / us / 1e6:9.1f} TF")
        lib.slam_set_option(None, b"gemm_tn224", 1)
    x, w, y = rb(M, 8192), rb(8192, 8192), torch.empty(M, 8192, dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, 8192, 8192, 1, st))
    print(f"nt 8192^3 (256x256 kernel) {us:9.1f} us {2.0 * M * 8192 * 8192 / us / 1e6:9.1f} TF")
