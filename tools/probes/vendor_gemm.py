"""External ceiling for the GEMMs: the vendor library (hipBLASLt / rocBLAS behind torch.matmul, bf16)
on the same box, next to the engine's own kernels on the same operands.

TOOLS ONLY: nothing under slamkit_amd/ and nothing inside bench.py's timed region imports or calls this; torch.matmul is used
here purely as an independent implementation to calibrate what the part sustains.

  python tools/probes/vendor_gemm.py [out.md]

Shapes: 8192^3 and the five matmul shapes of the Slam-358M step at M = 8192 tokens (NT forward / dgrad forms and the two
big weight-gradient TN forms). Operands: all-zero (no datapath toggling) vs N(0, 1) random - the protocol of
tools/probes/power_probe.py: 5 warm-up + 20 timed back-to-back launches per row, HIP events on the launch stream, clocks and
power read from rocm-smi right after."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from slamkit_amd import engine as E  # noqa: E402

lib = E.load_library()
st = E.current_stream_ptr()
dev = "cuda"
PEAK = 2.5e15


def timed(f, n=20, warm=5):
    for _ in range(warm):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def operands(kind, *shape):
    if kind == "zeros":
        return torch.zeros(*shape, device=dev, dtype=torch.bfloat16)
    return torch.randn(*shape, device=dev).to(torch.bfloat16)


rows = []
# (label, form, M, N, K): NT  Y[M,N] = X[M,K] W[N,K]^T ;  TN  dW[N,K] = dY[M,N]^T X[M,K] (contraction over M)
SHAPES = [("square 8192^3", "NT", 8192, 8192, 8192),
          ("gate|up fwd", "NT", 8192, 9728, 896),
          ("down fwd / gate|up dgrad", "NT", 8192, 896, 4864),
          ("gate|up dgrad (K 9728)", "NT", 8192, 896, 9728),
          ("qkv fwd", "NT", 8192, 1152, 896),
          ("o fwd", "NT", 8192, 896, 896),
          ("gate|up wgrad", "TN", 8192, 9728, 896),
          ("down wgrad", "TN", 8192, 896, 4864)]
for label, form, M, N, K in SHAPES:
    for kind in ("zeros", "randn"):
        flops = 2.0 * M * N * K
        if form == "NT":
            X, W = operands(kind, M, K), operands(kind, N, K)
            Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            us_v = timed(lambda: torch.matmul(X, W.t(), out=Y))
            us_e = timed(lambda: lib.slam_op_gemm_nt(X.data_ptr(), W.data_ptr(), Y.data_ptr(), None, None, M, N, K, 1, st))
        else:
            dY, X = operands(kind, M, N), operands(kind, M, K)
            dW32 = torch.empty(N, K, device=dev, dtype=torch.float32)
            dWb = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
            ws = torch.empty(lib.slam_op_gemm_tn_workspace(M, N, K) // 4 + 16, device=dev, dtype=torch.float32)
            us_v = timed(lambda: torch.matmul(dY.t(), X, out=dWb))   # vendor: bf16 output (its fp32-out form is not exposed by torch)
            us_e = timed(lambda: lib.slam_op_gemm_tn(dY.data_ptr(), X.data_ptr(), dW32.data_ptr(), 0, M, N, K, ws.data_ptr(), st))
        rows.append((label, form, M, N, K, kind, us_v, flops / us_v / 1e6, us_e, flops / us_e / 1e6))
        print(f"{label:28s} {form} {kind:6s} vendor {us_v:8.1f} us {flops / us_v / 1e6:7.1f} TF/s | engine {us_e:8.1f} us {flops / us_e / 1e6:7.1f} TF/s", flush=True)

try:
    smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout[-1200:]
except Exception as e:  # noqa: BLE001
    smi = f"rocm-smi: {e}"
print(smi)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("# Vendor-library GEMM probe (hipBLASLt / rocBLAS through torch.matmul, bf16) beside the engine's kernels\n\n")
        f.write(f"torch {torch.__version__}, {torch.cuda.get_device_name(0)}; 5 warm-up + 20 timed launches per row, HIP events; "
                "TFLOP/s against the 2.5 PFLOP/s dense bf16 peak in brackets. Tools only (never on the product path).\n\n")
        f.write("| shape | form | M | N | K | operands | vendor us | vendor TFLOP/s | engine us | engine TFLOP/s |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for label, form, M, N, K, kind, uv, tv, ue, te in rows:
            f.write(f"| {label} | {form} | {M} | {N} | {K} | {kind} | {uv:.1f} | {tv:.0f} ({tv * 1e12 / PEAK:.2f}) | {ue:.1f} | {te:.0f} ({te * 1e12 / PEAK:.2f}) |\n")
        f.write("\n```\n" + smi + "\n```\n")
