import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr()
M, N, K = 8192, 9728, 896
x = (torch.randn(M, K, device="cuda") * .5).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * .02).to(torch.bfloat16)
y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); a = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for rep in range(2):
  for stg in (0, 1, 2):
    nt = stg
    lib.slam_set_option(None, b"gemm_256_var", stg)
    f = timeit(lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), a.data_ptr(), M, N, K, st))
    p = timeit(lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 2, st))
    print(f"256-kernel variant {nt}: fused {f:7.1f} us  plain {p:7.1f} us")
