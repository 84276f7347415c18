"""Start offsets for EVERY persistent block of the 256 x 256 launches (gemm_256_cohorts: the slots of an XCD in NC contiguous
cohorts, cohort c starts c * stagger ticks late) against the round-3 rule (only the slots with a tile of slack start late).
Lockstep blocks store their tiles in one burst (50 MB per round) and wait for the store acknowledgements at the first
counted vmcnt of the next tile; offsets spread the bursts. HIP events, interleaved. Usage: python tools/probes/cohort_probe.py [iters]"""
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
cases = [("gate|up + SwiGLU", "swiglu", 8192, 9728, 896), ("gate|up plain", "plain", 8192, 9728, 896), ("down dgrad + dSwiGLU", "dswiglu", 8192, 4864, 896),
         ("qwen gate|up + SwiGLU", "swiglu", 16384, 17920, 1536), ("qwen down dgrad + dSwiGLU", "dswiglu", 16384, 8960, 1536)]
settings = [(0, 1200)] + [(nc, t) for nc in (2, 3, 4, 8) for t in (200, 400, 700, 1000)] + [(0, 1200)]
for name, kind, M, N, K in cases:
    x, w = rb(M, K), rb(N, K)
    y = torch.empty(M, N if kind != "dswiglu" else 2 * N, dtype=torch.bfloat16, device=dev)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev) if kind == "swiglu" else None
    if kind == "dswiglu": y.copy_(rb(M, 2 * N))
    fn = {"plain": lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 1, st),
          "swiglu": lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, N, K, st),
          "dswiglu": lambda: lib.slam_op_gemm_nt_dswiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, st)}[kind]
    out = []
    for nc, t in settings:
        lib.slam_set_option(None, b"gemm_256_cohorts", nc)
        lib.slam_set_option(None, b"gemm_256_stagger", t); lib.slam_set_option(None, b"gemm_256_stagger_dswiglu", t)
        out.append(min(timeit(fn), timeit(fn)))
    lib.slam_set_option(None, b"gemm_256_cohorts", 0)
    lib.slam_set_option(None, b"gemm_256_stagger", 1200); lib.slam_set_option(None, b"gemm_256_stagger_dswiglu", 1200)
    print(f"{name} {M}x{N}x{K}: " + "  ".join(f"[{nc}x{t}] {u:.1f}" for (nc, t), u in zip(settings, out)), flush=True)
    del x, w, y, act
