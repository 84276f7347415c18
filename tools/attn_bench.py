"""Attention micro-benchmark at the bench shape (8 x 1024 tokens, 14/2 heads): HIP-event timings of the
fwd and bwd op entry points. Usage: python tools/attn_bench.py [iters] [head_dim]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
HD = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B, T, nH, nKV = (8, 1024, 14, 2) if HD == 64 else (4, 2048, 12, 2)
M, ld = B * T, (nH + 2 * nKV) * HD
qkv = (torch.randn(M, ld, device="cuda")).to(torch.bfloat16)
o = torch.empty(M, nH * HD, dtype=torch.bfloat16, device="cuda")
do = torch.randn(M, nH * HD, device="cuda").to(torch.bfloat16)
dqkv = torch.empty(M, ld, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(nH * M, dtype=torch.float32, device="cuda")
ws = torch.empty(lib.slam_op_attn_bwd_workspace(M, nH, HD) // 4 + 16, dtype=torch.float32, device="cuda")
ss = (torch.arange(M, device="cuda", dtype=torch.int32) // T) * T
se = ss + T
def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
fl = 4 * HD * (T * (T + 1) / 2) * B * nH
f = timeit(lambda: lib.slam_op_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), ss.data_ptr(), M, nH, nKV, HD, st))
b = timeit(lambda: lib.slam_op_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), ws.data_ptr(), ss.data_ptr(), se.data_ptr(), M, nH, nKV, HD, st))
print(f"attn fwd {f:8.1f} us  {fl/f/1e6:7.1f} TF   bwd(all 4 kernels) {b:8.1f} us  {2.5*fl/b/1e6:7.1f} TF(5-matmul equiv)")
