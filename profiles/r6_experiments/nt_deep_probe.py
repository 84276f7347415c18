"""gemm_nt_deep A/B (round 6): the 128 x 128 NT DMA kernel on its 2-deep ring (two blocks per CU) vs a 3-deep ring with counted
vmcnt (one block per CU) on the step's 128 x 128 launches. HIP events, interleaved; outputs compared bit for bit.
Usage: python tools/probes/nt_deep_probe.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr(); dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
def rb(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn):
    for _ in range(4): assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 8192
for name, N, K, resid in (("qkv fwd", 1152, 896, False), ("o fwd", 896, 896, True), ("down fwd", 896, 4864, True), ("qkv dgrad", 896, 1152, False),
                          ("LM head", 512, 896, False)):
    x, w = rb(M, K), rb(N, K)
    r = rb(M, N) if resid else None
    ys = {}
    res = {0: [], 1: []}
    for _ in range(3):
        for deep in (0, 1):
            lib.slam_set_option(None, b"gemm_nt_deep", deep)
            y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
            fn = lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, r.data_ptr() if resid else None, M, N, K, 1, st)
            res[deep].append(timeit(fn))
            ys[deep] = y
    lib.slam_set_option(None, b"gemm_nt_deep", 0)
    same = bool(torch.equal(ys[0], ys[1]))
    print(f"{name:10s} {M}x{N}x{K}: 2-deep {min(res[0]):6.1f}/{max(res[0]):6.1f} us   3-deep {min(res[1]):6.1f}/{max(res[1]):6.1f} us   bit-identical {same}", flush=True)
