"""A/B of one 0 / 1 GEMM option over the persistent 256 x 256 launches of the Slam-358M step (and of the configs[3]-shaped model):
gemm_256_roles (loader / storer wave rows) or gemm_256_batch_loads (SwiGLU-backward epilogue with its gate|up loads batched).
HIP events, interleaved, N(0, 1)-scale operands. Usage: python tools/probes/roles_probe.py [iters] [option]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr(); dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
OPT = (sys.argv[2] if len(sys.argv) > 2 else "gemm_256_roles").encode()
DEFAULT = {b"gemm_256_roles": 0, b"gemm_256_batch_loads": 1}[OPT]
def rb(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn):
    for _ in range(4): assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cases = [("gate|up + SwiGLU", "swiglu", 8192, 9728, 896), ("gate|up plain", "plain", 8192, 9728, 896), ("down dgrad + dSwiGLU", "dswiglu", 8192, 4864, 896),
         ("LM head 152k", "plain", 16384, 152320, 1536), ("gate|up + SwiGLU 1.5B", "swiglu", 16384, 17920, 1536), ("down dgrad + dSwiGLU 1.5B", "dswiglu", 16384, 8960, 1536)]
print(f"{'case':44s} {OPT.decode() + ' = 0':>24s} {OPT.decode() + ' = 1':>24s}")
for name, kind, M, N, K in cases:
    x, w = rb(M, K), rb(N, K)
    y = torch.empty(M, N if kind != "dswiglu" else 2 * N, dtype=torch.bfloat16, device=dev)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev) if kind == "swiglu" else None
    if kind == "dswiglu": y.copy_(rb(M, 2 * N))
    fn = {"plain": lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 1, st),
          "swiglu": lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, N, K, st),
          "dswiglu": lambda: lib.slam_op_gemm_nt_dswiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, st)}[kind]
    r = {0: [], 1: []}
    for _ in range(3):
        for roles in (0, 1):
            lib.slam_set_option(None, OPT, roles)
            r[roles].append(timeit(fn))
    lib.slam_set_option(None, OPT, DEFAULT)
    print(f"{name + f' {M}x{N}x{K}':44s} " + " ".join(f"{min(r[k]):11.1f}/{max(r[k]):11.1f}" for k in (0, 1)), flush=True)
    del x, w, y, act
