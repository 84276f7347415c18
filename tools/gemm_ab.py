"""Same-process A/B of the NT GEMM variants on the Slam-358M shapes (run on the GPU box):
  python tools/gemm_ab.py [modes...]      modes = gemm_glds values (2 = default ring, 11, 3, 4, 322-325, 82-84, 162/163)
  env CMODES / N112 / G256 = comma lists swept for mode 2 (column-tile layout, 128x112 tiles, 256x256 kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library()
M = 8192; dev = "cuda"; st = E.current_stream_ptr()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def rb(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
modes = [int(a) for a in sys.argv[1:]] or [2, 11]
cmodes = [int(c) for c in os.environ.get("CMODES", "1").split(",")]
n112s = [int(c) for c in os.environ.get("N112", "0").split(",")]
g256s = [int(c) for c in os.environ.get("G256", "0").split(",")]
ref = {}
def run(name, N, K, mode, cm, n112, g256):
    x, w = data[name]
    y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, mode, st))
    if name not in ref: ref[name] = y
    ok = torch.equal(ref[name], y)
    print(f"nt {name:16s} mode {mode:4d} cmode {cm} n112 {n112} g256 {g256} {us:8.1f} us {2.0*M*N*K/us/1e6:8.1f} TF  same={ok}", flush=True)
data = {}
for name, (N, K) in {"qkv fwd": (1152, 896), "o fwd": (896, 896), "gate_up fwd": (9728, 896), "down fwd": (896, 4864),
                     "down dgrad": (4864, 896), "gate_up dgrad": (896, 9728)}.items():
    data[name] = (rb(M, K), rb(N, K))
    for mode in modes:
      for cm in (cmodes if mode == 2 else [0]):
       for n112 in (n112s if mode == 2 else [0]):
        for g256 in (g256s if mode == 2 else [0]):
         lib.slam_set_option(None, b"gemm_cmode", cm)
         lib.slam_set_option(None, b"gemm_n112", n112)
         lib.slam_set_option(None, b"gemm_256", g256)
         run(name, N, K, mode, cm, n112, g256)
