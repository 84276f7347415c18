"""Prefetch-depth probe of the 8-phase GEMM kernels: the same launches on libraries built with a stricter steady-state
vmcnt (SLAM_PROBE_VMCNT = 4 (default: two 16 KB half-tiles beyond the awaited one stay in flight), 2 (one), 0 (none)).
If the main loops were MFMA- or LDS-bound the count would not matter; a time that grows as the depth shrinks says they
are bound by bytes in flight x memory latency. Run: SLAM_ENGINE_LIB=<lib> python tools/probes/depth_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slamkit_amd import engine as E

lib = E.load_library()
st = E.current_stream_ptr()
dev = "cuda"


def timed(f, n=30, warm=8):
    for _ in range(warm):
        assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


bf = lambda *s, sc=0.5: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)  # noqa: E731
tag = os.path.basename(os.environ.get("SLAM_ENGINE_LIB", "default"))
for (M, N, K, kind) in ((8192, 9728, 896, "swiglu"), (8192, 8192, 8192, "plain"), (8192, 896, 9728, "nt224"), (8192, 4864, 896, "plain")):
    X, W = bf(M, K), bf(N, K, sc=0.05)
    Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if kind == "swiglu":
        A = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
        f = lambda: lib.slam_op_gemm_nt_swiglu(X.data_ptr(), W.data_ptr(), Y.data_ptr(), A.data_ptr(), M, N, K, st)
    else:
        if kind == "nt224":
            lib.slam_set_option(None, b"gemm_nt224", 2)
        f = lambda: lib.slam_op_gemm_nt(X.data_ptr(), W.data_ptr(), Y.data_ptr(), None, None, M, N, K, 1, st)
    us = timed(f)
    lib.slam_set_option(None, b"gemm_nt224", 1)
    print(f"{tag:28s} {kind:7s} M{M} N{N} K{K}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)
