"""gemm_256_persist A/B on the Slam-358M / Qwen2.5-1.5B-shaped launches that the persistent 256 x 256 kernel can serve:
HIP-event times, interleaved. Usage: python tools/persist_probe.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr(); dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
def rb(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cases = [("gate|up fwd + SwiGLU  8192x9728x896", "swiglu", 8192, 9728, 896), ("gate|up fwd plain     8192x9728x896", "plain", 8192, 9728, 896),
         ("down dgrad + dSwiGLU  8192x4864x896", "dswiglu", 8192, 4864, 896),
         ("qwen gate|up + SwiGLU 16384x17920x1536", "swiglu", 16384, 17920, 1536), ("qwen down dgrad      16384x8960x1536", "dswiglu", 16384, 8960, 1536),
         ("qwen LM head          16384x152320x1536", "plain", 16384, 152320, 1536)]
for name, kind, M, N, K in cases:
    x, w = rb(M, K), rb(N, K)
    y = torch.empty(M, N if kind != "dswiglu" else 2 * N, dtype=torch.bfloat16, device=dev)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev) if kind == "swiglu" else None
    if kind == "dswiglu": y.copy_(rb(M, 2 * N))
    fn = {"plain": lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 2, st),
          "swiglu": lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, N, K, st),
          "dswiglu": lambda: lib.slam_op_gemm_nt_dswiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, st)}[kind]
    res = []
    for persist in (0, 1, 0, 1):
        lib.slam_set_option(None, b"gemm_256_persist", persist)
        res.append(timeit(fn))
    lib.slam_set_option(None, b"gemm_256_persist", 1)
    fl = 2.0 * M * N * K
    print(f"{name:42s} one block per tile {res[0]:7.1f} / {res[2]:7.1f} us   persistent {res[1]:7.1f} / {res[3]:7.1f} us   "
          f"({fl / min(res[0], res[2]) / 1e6:6.0f} -> {fl / min(res[1], res[3]) / 1e6:6.0f} TFLOP/s)", flush=True)
    del x, w, y, act
