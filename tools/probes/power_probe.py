"""Is the long-K GEMM rate a kernel limit or a clock/power limit? The same launches on zero operands (no datapath
toggling), small-magnitude and full-range random operands, for both 256x256 kernels. Also reads the clock the SMI reports."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slamkit_amd import engine as E

lib = E.load_library()
st = E.current_stream_ptr()
dev = "cuda"


def timed(f, n=20, warm=5):
    for _ in range(warm):
        assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


M = N = K = 8192
Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for name, mk in (("zeros", lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16)),
                 ("ones", lambda *s: torch.ones(*s, device=dev, dtype=torch.bfloat16)),
                 ("randn*0.05", lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)),
                 ("randn", lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)),
                 ("uniform bits", lambda *s: torch.randint(-32768, 32767, s, device=dev, dtype=torch.int16).view(torch.bfloat16).nan_to_num(0.0, 1.0, -1.0))):
    X, W = mk(M, K), mk(N, K)
    for w4 in (0, 1):
        lib.slam_set_option(None, b"gemm_256_w4", w4)
        us = timed(lambda: lib.slam_op_gemm_nt(X.data_ptr(), W.data_ptr(), Y.data_ptr(), None, None, M, N, K, 1, st))
        print(f"{name:14s} w4={w4}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)
lib.slam_set_option(None, b"gemm_256_w4", 0)
try:
    print(subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as e:  # noqa: BLE001
    print("rocm-smi:", e)
