"""Late start of the persistent 256x256 blocks that have a tile of slack (gemm_256_stagger / gemm_256_stagger_dswiglu, in
10-ns ticks): stand-alone launch times of the gate|up forward (+SwiGLU) and the down-proj dgrad (+SwiGLU backward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slamkit_amd import engine as E

lib = E.load_library()
st = E.current_stream_ptr()
dev = "cuda"
M, H, I = 8192, 896, 4864
bf = lambda *s, sc=0.5: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)  # noqa: E731


def timed(f, n=30, warm=10):
    for _ in range(warm):
        assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


x, wgu = bf(M, H), bf(2 * I, H, sc=0.02)
gu, act = torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16), torch.empty(M, I, device=dev, dtype=torch.bfloat16)
dy, wdt = bf(M, H, sc=0.1), bf(I, H, sc=0.02)
gu2 = bf(M, 2 * I)
for ticks in (0, 400, 800, 1200, 1600, 2000, 2600, 0):
    lib.slam_set_option(None, b"gemm_256_stagger", ticks)
    lib.slam_set_option(None, b"gemm_256_stagger_dswiglu", ticks)
    a = timed(lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), wgu.data_ptr(), gu.data_ptr(), act.data_ptr(), M, 2 * I, H, st))
    b = timed(lambda: lib.slam_op_gemm_nt_dswiglu(dy.data_ptr(), wdt.data_ptr(), gu2.data_ptr(), M, I, H, st))
    print(f"stagger {ticks / 100:5.1f} us: gate|up fwd + SwiGLU {a:7.1f} us   down dgrad + dSwiGLU {b:7.1f} us", flush=True)
lib.slam_set_option(None, b"gemm_256_stagger", 0); lib.slam_set_option(None, b"gemm_256_stagger_dswiglu", 0)
