"""Attention micro-benchmark: HIP-event timings of the fwd and bwd op entry points over shapes, libraries and tunes.

Usage: python tools/attn_bench.py [--iters N] [--hd 64|128] [--shapes 8x1024,1x8192,...] [--libs new,r2]
                                  [--tunes jq.kw.nch,...]      (new library only)
The r2 library (slamkit_amd/lib/libslam_engine_r2.so, if present) is the round-2 build kept for same-box A/B.
"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--hd", type=int, default=64)
ap.add_argument("--shapes", default="8x1024")
ap.add_argument("--libs", default="new")
ap.add_argument("--tunes", default="1.1.4")
ap.add_argument("--heads", default="14,2")
a = ap.parse_args()
HD = a.hd
nH, nKV = [int(x) for x in a.heads.split(",")]
libdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slamkit_amd", "lib")
st = E.current_stream_ptr()


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for shape in a.shapes.split(","):
    B, T = [int(x) for x in shape.split("x")]
    M, ld = B * T, (nH + 2 * nKV) * HD
    torch.manual_seed(0)
    qkv = torch.randn(M, ld, device="cuda").to(torch.bfloat16)
    o = torch.empty(M, nH * HD, dtype=torch.bfloat16, device="cuda")
    do = torch.randn(M, nH * HD, device="cuda").to(torch.bfloat16)
    dqkv = torch.empty(M, ld, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(nH * M, dtype=torch.float32, device="cuda")
    ss = (torch.arange(M, device="cuda", dtype=torch.int32) // T) * T
    se = ss + T
    fl = 4 * HD * (T * (T + 1) / 2) * B * nH
    ref = {}
    for libname in a.libs.split(","):
        path = None if libname == "new" else os.path.join(libdir, f"libslam_engine_{libname}.so")
        if path and not os.path.exists(path):
            print(f"{libname}: {path} missing, skipped")
            continue
        lib = E.load_library(path)
        ws = torch.empty(lib.slam_op_attn_bwd_workspace(M, nH, HD) // 4 + 16, dtype=torch.float32, device="cuda")
        tunes = a.tunes.split(",") if libname == "new" else ["-"]
        for tune in tunes:
            if tune != "-":
                jq, kw, nch, prio = ([int(x) for x in tune.split(".")] + [0])[:4]
                for k, v in (("attn_jq", jq), ("attn_kw", kw), ("attn_nch", nch), ("attn_prio", prio)):
                    assert lib.slam_set_option(None, k.encode(), v) == 0
            f = timeit(lambda: lib.slam_op_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), ss.data_ptr(), M, nH, nKV, HD, st), a.iters)
            b = timeit(lambda: lib.slam_op_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), ws.data_ptr(), ss.data_ptr(), se.data_ptr(), M, nH, nKV, HD, st), a.iters)
            torch.cuda.synchronize()
            key = "o+dqkv"
            cur = (o.float().clone(), dqkv.float().clone())
            dev = ""
            if key in ref:
                dev = "  vs first: o %.2e dqkv %.2e (rel rms)" % tuple(
                    float((x - y).pow(2).mean().sqrt() / (y.pow(2).mean().sqrt() + 1e-30)) for x, y in zip(cur, ref[key]))
            else:
                ref[key] = cur
            print(f"B{B} T{T} hd{HD} {libname:4s} tune {tune:7s} fwd {f:7.1f} us {fl/f/1e6:7.1f} TF | bwd (plan+dq+dkv+reduce) {b:7.1f} us "
                  f"{2.5*fl/b/1e6:7.1f} TF(5-matmul equiv){dev}", flush=True)
