"""What do the output stores of the K = 896 launches cost? The persistent 256 x 256 launches with and without their stores
(gemm_nt_store = 2: probe-only switch, the accumulators are still converted), and at K = 128 (two K-tiles per tile: a launch
that is almost only epilogue). HIP events, interleaved, random operands. Usage: python tools/probes/epilogue_cost.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slamkit_amd import engine as E
# the no-store switch exists only in the -DSLAM_PROBES build (python -m slamkit_amd.csrc.build --probes); the product library rejects it
from slamkit_amd.csrc import build as _B
lib = E.load_library(_B.build(verbose=False, probes=True)); st = E.current_stream_ptr(); dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
def rb(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn):
    for _ in range(4): assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cases = [("gate|up + SwiGLU", "swiglu", 8192, 9728, 896), ("gate|up plain", "plain", 8192, 9728, 896), ("down dgrad + dSwiGLU", "dswiglu", 8192, 4864, 896),
         ("gate|up + SwiGLU K=128", "swiglu", 8192, 9728, 128), ("gate|up plain K=128", "plain", 8192, 9728, 128), ("dSwiGLU K=128", "dswiglu", 8192, 4864, 128),
         ("gate|up + SwiGLU K=1792", "swiglu", 8192, 9728, 1792)]
print(f"{'case':40s} {'stores':>16s} {'no stores':>16s} {'stagger 0':>16s}")
for name, kind, M, N, K in cases:
    x, w = rb(M, K), rb(N, K)
    y = torch.empty(M, N if kind != "dswiglu" else 2 * N, dtype=torch.bfloat16, device=dev)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev) if kind == "swiglu" else None
    if kind == "dswiglu": y.copy_(rb(M, 2 * N))
    fn = {"plain": lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 1, st),
          "swiglu": lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, N, K, st),
          "dswiglu": lambda: lib.slam_op_gemm_nt_dswiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, st)}[kind]
    r = {0: [], 2: [], "s0": []}
    for _ in range(2):
        for ns in (0, 2):
            lib.slam_set_option(None, b"gemm_nt_store", ns)
            r[ns].append(timeit(fn))
        lib.slam_set_option(None, b"gemm_nt_store", 0)
        lib.slam_set_option(None, b"gemm_256_stagger", 0); lib.slam_set_option(None, b"gemm_256_stagger_dswiglu", 0)
        r["s0"].append(timeit(fn))
        lib.slam_set_option(None, b"gemm_256_stagger", 1200); lib.slam_set_option(None, b"gemm_256_stagger_dswiglu", 1200)
    print(f"{name + f' {M}x{N}x{K}':40s} " + " ".join(f"{min(r[k]):7.1f}/{max(r[k]):7.1f}" for k in (0, 2, "s0")), flush=True)
