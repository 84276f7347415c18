"""The persistent 256 x 256 launches with and without their output stores (gemm_nt_store = 2, probe only) under ONE rocprofv3 counter
pass: cycles vs wall -> do the stores cost cycles (stall) or clock (power)? Summary: tools/probes/power_or_stall_summary.py."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slamkit_amd import engine as E
# the no-store switch exists only in the -DSLAM_PROBES build (python -m slamkit_amd.csrc.build --probes); the product library rejects it
from slamkit_amd.csrc import build as _B
lib = E.load_library(_B.build(verbose=False, probes=True)); st = E.current_stream_ptr(); dev = "cuda"
def rb(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
manifest = []
for name, kind, M, N, K in [("gate|up + SwiGLU", "swiglu", 8192, 9728, 896), ("gate|up plain", "plain", 8192, 9728, 896), ("down dgrad + dSwiGLU", "dswiglu", 8192, 4864, 896)]:
    x, w = rb(M, K), rb(N, K)
    y = torch.empty(M, N if kind != "dswiglu" else 2 * N, dtype=torch.bfloat16, device=dev)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev) if kind == "swiglu" else None
    if kind == "dswiglu": y.copy_(rb(M, 2 * N))
    fn = {"plain": lambda: lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 1, st),
          "swiglu": lambda: lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, N, K, st),
          "dswiglu": lambda: lib.slam_op_gemm_nt_dswiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, st)}[kind]
    torch.cuda.synchronize()
    for ns in (0, 2):
        lib.slam_set_option(None, b"gemm_nt_store", ns)
        for _ in range(8): assert fn() == 0
        torch.cuda.synchronize()
        manifest.append({"shape": name, "M": M, "N": N, "K": K, "fill": "randn", "variant": "stores" if ns == 0 else "no stores", "launches": 8})
lib.slam_set_option(None, b"gemm_nt_store", 0)
json.dump(manifest, open(sys.argv[1], "w"))
