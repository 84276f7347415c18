"""Would a 2-way split-K of the N = 896 / long-K launches pay? (round 6 probe, tools only.) The down projection forward
[8192 x 4864] x [4864 -> 896] runs on 448 tiles of the 128 x 128 kernel (84 us in the step); the 256 x 224 kernel has 128 tiles - half the
chip. Emulation of a split-K pair with the kernels that exist: the two K-halves as two concurrent launches on two streams (128 blocks
each), against the single launches. HIP events on the first stream around the pair (the second stream joined by events).
Usage: python tools/probes/splitk_down_probe.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slamkit_amd import engine as E
lib = E.load_library(); dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
def rb(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, M, N, K in (("down fwd", 8192, 896, 4864), ("gate|up dgrad", 8192, 896, 9728)):
    x, w = rb(M, K), rb(N, K)
    xa, xb, wa, wb = x[:, :K // 2].contiguous(), x[:, K // 2:].contiguous(), w[:, :K // 2].contiguous(), w[:, K // 2:].contiguous()
    y, ya, yb = (torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(3))
    def single(nt224):
        lib.slam_set_option(None, b"gemm_nt224", nt224)
        return lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 1, s1.cuda_stream)
    def pair():
        lib.slam_set_option(None, b"gemm_nt224", 2)
        e = torch.cuda.Event(); e.record(s1); s2.wait_event(e)
        r = lib.slam_op_gemm_nt(xa.data_ptr(), wa.data_ptr(), ya.data_ptr(), None, None, M, N, K // 2, 1, s1.cuda_stream)
        r |= lib.slam_op_gemm_nt(xb.data_ptr(), wb.data_ptr(), yb.data_ptr(), None, None, M, N, K // 2, 1, s2.cuda_stream)
        e2 = torch.cuda.Event(); e2.record(s2); s1.wait_event(e2)
        return r
    def timeit(fn):
        for _ in range(4): assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(iters): fn()
        e1.record(s1); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    res = {}
    for _ in range(2):
        for k, fn in (("128x128 kernel (448 tiles)", lambda: single(0)), ("256x224 kernel (128 tiles)", lambda: single(2)),
                      ("two K-halves, 256x224, two streams (2 x 128 tiles)", pair)):
            res.setdefault(k, []).append(timeit(fn))
    lib.slam_set_option(None, b"gemm_nt224", 1)
    print(f"{name} {M}x{N}x{K}: " + "; ".join(f"{k}: {min(v):.1f}/{max(v):.1f} us" for k, v in res.items()), flush=True)
