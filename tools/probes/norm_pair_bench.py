"""Why the RMSNorm backward takes 2.3x (H 896) / 5.7x (H 1536) its stand-alone time inside the step.

The kernel is timed alone and BESIDE the weight-gradient GEMM that occupies the engine's side stream at that point of the
backward (the 256 x 224 kernel, one 512-thread block per tile: 128 KB of LDS and 2 x 224 VGPRs per SIMD on every CU it sits
on), launched on a second stream right before it - both kernel variants, a sweep of the lean kernel's grid:
  lean = 0: the register-pipelined kernel (144-200 VGPRs, 16-24 KB LDS): cannot share a CU with a GEMM block;
  lean = 1: the <= 64-VGPR, < 7 KB-LDS kernel: one wave per SIMD fits beside a GEMM block.
  python tools/probes/norm_pair_bench.py [out.md]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from slamkit_amd import engine as E  # noqa: E402

lib = E.load_library()
dev = "cuda"
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
out_lines = []


def say(x):
    print(x, flush=True)
    out_lines.append(x)


def case(M, H, bgN, bgK, label):
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731
    x, dy, dres, w = bf(M, H), bf(M, H), bf(M, H), bf(H)
    dx = torch.empty_like(x)
    rstd = torch.rand(M, device=dev) + 0.5
    ws = torch.empty(lib.slam_op_rmsnorm_bwd_workspace(M, H) // 4 + 16, dtype=torch.float32, device=dev)
    # background: dW[bgN][bgK] = dY[M][bgN]^T X[M][bgK] as the step launches it (256 x 224 tiles, no K-split)
    dY, X = bf(M, bgN), bf(M, bgK)
    dW = torch.empty(bgN, bgK, dtype=torch.float32, device=dev)
    gws = torch.empty(lib.slam_op_gemm_tn_workspace(M, bgN, bgK) // 4 + 16, dtype=torch.float32, device=dev)
    for k, v in ((b"gemm_tn224", 2), (b"gemm_tn224_max_split", 1)):
        assert lib.slam_set_option(None, k, v) == 0
    p = lambda t: t.data_ptr()  # noqa: E731

    def norm():
        return lib.slam_op_rmsnorm_bwd(p(dy), p(x), p(w), p(rstd), p(dres), p(dx), None, p(ws), M, H, s_main.cuda_stream)

    def bg():
        return lib.slam_op_gemm_tn(p(dY), p(X), p(dW), 0, M, bgN, bgK, p(gws), s_side.cuda_stream)

    def timed(with_bg, reps=15):
        ts, tb = [], []
        for _ in range(reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if with_bg:
                b0.record(s_side)
                assert bg() == 0
                b1.record(s_side)
                torch.cuda._sleep(20000)  # ~10 us of the default stream: lets the background blocks become resident first
                s_main.wait_stream(torch.cuda.current_stream())
            e0.record(s_main)
            assert norm() == 0
            e1.record(s_main)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
            if with_bg:
                tb.append(b0.elapsed_time(b1) * 1e3)
        ts.sort(); tb.sort()
        return ts[len(ts) // 2], (tb[len(tb) // 2] if tb else 0.0)

    nbytes = 8 * M * H
    say(f"\n### {label}: rmsnorm_bwd [{M} x {H}] (+ residual add, {nbytes / 1e6:.1f} MB) beside dW[{bgN}][{bgK}] over {M} tokens "
        f"({(bgN // 256) * (bgK // 224) if bgN % 256 == 0 and bgK % 224 == 0 else (bgK // 256) * (bgN // 224)} one-per-CU blocks)\n")
    say("| kernel | grid option | alone us | GB/s alone | beside the GEMM us | slow-down | GEMM us (with the norm beside it) |")
    say("|---|---|---|---|---|---|---|")
    for lean, blocks in ((0, 0), (2, 512), (2, 1024), (2, 2048)):
        lib.slam_set_option(None, b"norm_bwd_lean", lean)
        if blocks:
            lib.slam_set_option(None, b"norm_bwd_blocks", blocks)
        a, _ = timed(False)
        b, g = timed(True)
        say(f"| {'lean (<= 64 VGPR)' if lean else 'register-pipelined (144+ VGPR)'} | {blocks or 'min(M/16, 512)'} | {a:.1f} | {nbytes / a / 1e3:.0f} | {b:.1f} | {b / a:.2f}x | {g:.1f} |")
    lib.slam_set_option(None, b"norm_bwd_lean", 0)
    lib.slam_set_option(None, b"norm_bwd_blocks", 2048)
    lib.slam_set_option(None, b"gemm_tn224", 1)
    lib.slam_set_option(None, b"gemm_tn224_max_split", 16)


say("# RMSNorm backward alone and beside the side stream's weight-gradient GEMM (tools/probes/norm_pair_bench.py)")
case(16384, 1536, 1536, 8960, "configs[3]-shaped, down weight gradient (240 blocks)")
case(16384, 1536, 17920, 1536, "configs[3]-shaped, gate|up weight gradient (480 blocks, two rounds)")
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(out_lines) + "\n")
