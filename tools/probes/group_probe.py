"""2-D tile groups of the persistent 256 x 256 launches (gemm_group_rows_256 x gemm_group_cols_256): an XCD's contiguous range of tile
ids then covers GR row panels x GC column panels instead of GR x ALL column panels - fewer weight bytes through each XCD-private L2.
HIP events, interleaved, random operands. Usage: python tools/probes/group_probe.py [iters]"""
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
cases = [("gate|up + SwiGLU", "swiglu", 8192, 9728, 896, [(4, 0), (8, 19), (4, 19), (8, 10), (16, 10), (8, 0), (4, 0)]),
         ("gate|up plain", "plain", 8192, 9728, 896, [(4, 0), (8, 19), (4, 19), (8, 10), (16, 10), (8, 0), (4, 0)]),
         ("down dgrad + dSwiGLU", "dswiglu", 8192, 4864, 896, [(4, 0), (8, 10), (2, 0), (8, 0), (4, 0)]),
         ("qwen gate|up + SwiGLU", "swiglu", 16384, 17920, 1536, [(4, 0), (16, 35), (8, 35), (8, 0), (4, 0)]),
         ("qwen LM head", "plain", 16384, 152320, 1536, [(4, 0), (8, 0), (16, 75), (8, 149), (4, 0)])]
for name, kind, M, N, K, settings in cases:
    x, w = rb(M, K), rb(N, K)
    y = torch.empty(M, N if kind != "dswiglu" else 2 * N, dtype=torch.bfloat16, device=dev)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev) if kind == "swiglu" else None
    if kind == "dswiglu": y.copy_(rb(M, 2 * N))
    fn = {"plain": lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 1, st),
          "swiglu": lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, N, K, st),
          "dswiglu": lambda: lib.slam_op_gemm_nt_dswiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, st)}[kind]
    out = []
    for gr, gc in settings:
        lib.slam_set_option(None, b"gemm_group_rows_256", gr); lib.slam_set_option(None, b"gemm_group_cols_256", gc)
        out.append(min(timeit(fn), timeit(fn)))
    lib.slam_set_option(None, b"gemm_group_rows_256", 4); lib.slam_set_option(None, b"gemm_group_cols_256", 0)
    print(f"{name} {M}x{N}x{K} ({M // 256} x {N // 256} tiles): " + "  ".join(f"[{gr}x{gc or 'all'}] {u:.1f}" for (gr, gc), u in zip(settings, out)), flush=True)
    del x, w, y, act
