"""Power or stall? The gate|up projection launch (M 8192, N 9728, K 896) and the square
8192^3 launch on all-zero and on N(0,1) operands, for every main-loop variant of the 256 x 256 kernels, inside ONE process -
run it under ONE rocprofv3 counter pass:
  cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
      SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d <out> -o p -- python tools/probes/power_or_stall.py <manifest.json>
tools/probes/power_or_stall_summary.py turns the counter CSV + the manifest into the table (cycles, wall, effective clock =
cycles / wall, MFMA-busy). Same cycles and a different wall -> clock (power); fewer cycles on zeros at equal MFMA-busy -> issue
throttling; neither -> schedule."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slamkit_amd import engine as E  # noqa: E402

lib = E.load_library()
st = E.current_stream_ptr()
dev = "cuda"
LAUNCHES = int(os.environ.get("POS_LAUNCHES", "8"))
variants = [("16x16x32 8-wave", {"gemm_mf32": 0, "gemm_256_w4": 0}), ("32x32x16 8-wave", {"gemm_mf32": 1, "gemm_256_w4": 0}),
            ("32x32x16 4-wave", {"gemm_mf32": 1, "gemm_256_w4": 1})]
shapes = [("gate|up fwd", 8192, 9728, 896), ("square", 8192, 8192, 8192)]
fills = [("zeros", lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16)),
         ("randn", lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16))]
manifest = []
for sname, M, N, K in shapes:
    Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for fname, mk in fills:
        X, W = mk(M, K), mk(N, K)
        torch.cuda.synchronize()
        for vname, opts in variants:
            for k, v in opts.items():
                assert lib.slam_set_option(None, k.encode(), v) == 0
            for _ in range(LAUNCHES):
                assert lib.slam_op_gemm_nt(X.data_ptr(), W.data_ptr(), Y.data_ptr(), None, None, M, N, K, 1, st) == 0
            torch.cuda.synchronize()
            manifest.append({"shape": sname, "M": M, "N": N, "K": K, "fill": fname, "variant": vname, "launches": LAUNCHES})
lib.slam_set_option(None, b"gemm_mf32", 0)
lib.slam_set_option(None, b"gemm_256_w4", 0)
json.dump(manifest, open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/pos_manifest.json", "w"))
print("done", len(manifest), "groups")
