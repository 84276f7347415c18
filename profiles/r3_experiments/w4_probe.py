"""Four-wave 256x256 NT kernel (gemm_256_w4=1) against the eight-wave 8-phase kernel: same bits, launch time."""
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
for (M, N, K, kind) in ((8192, 9728, 896, "swiglu"), (8192, 9728, 896, "plain"), (8192, 8192, 8192, "plain"), (8192, 4864, 896, "plain"), (2048, 2048, 256, "plain")):
    X, W = bf(M, K), bf(N, K, sc=0.05)
    outs = {}
    for w4 in (0, 1):
        lib.slam_set_option(None, b"gemm_256_w4", w4)
        lib.slam_set_option(None, b"gemm_256", 2)
        Y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        if kind == "swiglu":
            A = torch.full((M, N // 2), float("nan"), device=dev, dtype=torch.bfloat16)
            f = lambda: lib.slam_op_gemm_nt_swiglu(X.data_ptr(), W.data_ptr(), Y.data_ptr(), A.data_ptr(), M, N, K, st)
        else:
            A = None
            f = lambda: lib.slam_op_gemm_nt(X.data_ptr(), W.data_ptr(), Y.data_ptr(), None, None, M, N, K, 1, st)
        us = timed(f)
        outs[w4] = (Y, A)
        print(f"w4={w4} {kind:7s} M{M} N{N} K{K}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)
    lib.slam_set_option(None, b"gemm_256_w4", 0); lib.slam_set_option(None, b"gemm_256", 1)
    same = torch.equal(outs[0][0], outs[1][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[1][1]))
    ref = (X[:64].float() @ W.float().t())
    err = (outs[1][0][:64].float() - ref).abs().max().item() if kind == "plain" else float("nan")
    print(f"   bit-identical to the eight-wave kernel: {same}; max |err| vs fp32 on 64 rows: {err:.4f}", flush=True)
