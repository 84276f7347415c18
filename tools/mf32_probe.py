"""Main-loop variants of the NT kernels, interleaved A/B with HIP events on the Slam-358M launches (random operands):
  16x16x32 (round 4) | 32x32x16 eight-wave | 32x32x16 four-wave persistent (256 x 256 tiles only)
Usage: python tools/mf32_probe.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
cases = [("gate|up fwd + SwiGLU", "swiglu", 8192, 9728, 896), ("gate|up fwd plain", "plain", 8192, 9728, 896),
         ("down dgrad + dSwiGLU", "dswiglu", 8192, 4864, 896), ("down fwd + resid", "resid", 8192, 896, 4864),
         ("o fwd + resid", "resid", 8192, 896, 896), ("qkv fwd (no rope)", "bias", 8192, 1152, 896), ("qkv dgrad", "plain", 8192, 896, 1152),
         ("head V=512", "plain", 8192, 512, 896), ("square 8192^3", "plain", 8192, 8192, 8192),
         ("qwen gate|up + SwiGLU", "swiglu", 16384, 17920, 1536), ("qwen LM head", "plain", 16384, 152320, 1536)]
variants = [("16x16x32", 0, 0), ("32x32x16 8w", 1, 0), ("32x32x16 4w", 1, 1)]
print(f"{'case':44s} " + " ".join(f"{v[0]:>22s}" for v in variants))
for name, kind, M, N, K in cases:
    x, w = rb(M, K), rb(N, K)
    y = torch.empty(M, N if kind != "dswiglu" else 2 * N, dtype=torch.bfloat16, device=dev)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev) if kind == "swiglu" else None
    res = rb(M, N) if kind == "resid" else None
    bias = rb(N) if kind == "bias" else None
    if kind == "dswiglu": y.copy_(rb(M, 2 * N))
    fn = {"plain": lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 1, st),
          "resid": lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, res.data_ptr(), M, N, K, 1, st),
          "bias": lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), bias.data_ptr(), None, M, N, K, 1, st),
          "swiglu": lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, N, K, st),
          "dswiglu": lambda: lib.slam_op_gemm_nt_dswiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, st)}[kind]
    res_us = {v[0]: [] for v in variants}
    for rnd_ in range(2):
        for vname, mf, w4 in variants:
            lib.slam_set_option(None, b"gemm_mf32", mf); lib.slam_set_option(None, b"gemm_256_w4", w4)
            res_us[vname].append(timeit(fn))
    lib.slam_set_option(None, b"gemm_mf32", 0); lib.slam_set_option(None, b"gemm_256_w4", 0)
    fl = 2.0 * M * N * K
    print(f"{name + f' {M}x{N}x{K}':44s} " + " ".join(f"{min(res_us[v[0]]):7.1f}/{max(res_us[v[0]]):7.1f} {fl / min(res_us[v[0]]) / 1e6:5.0f}T" for v in variants), flush=True)
    del x, w, y, act, res, bias
