"""Vocab-tiled LM head + cross-entropy vs the materialised [M][Vp] logits buffer, MEASURED (configs[3]:
M = 16,384 packed tokens, H = 1536, V = 152,167 -> Vp = 152,320).

Both pipelines are timed end to end with the engine's own kernels through the C ABI's single-op entry points:
  materialised (what the engine does): head GEMM -> [M][Vp] bf16 logits (5 GB) -> cross-entropy kernel rewrites them in place
      with d loss / d logits -> weight-gradient GEMM dE = dlogits^T hf -> dgrad GEMM dhf = dlogits E;
  tiled: for every panel of P vocabulary columns: head GEMM of the panel -> [M][P] buffer (P = 4096: 134 MB, inside the 256 MB
      infinity cache) -> row statistics pass over the panel; then, after the per-row log-sum-exp is known, per panel again:
      the SAME GEMM recomputed (no 5 GB buffer to keep it in) -> gradient pass over the panel -> panel weight-gradient GEMM ->
      panel dgrad GEMM accumulating into dhf (bf16 residual epilogue).
The tiled pipeline's two element-wise passes do not exist as kernels; each is timed as the engine's cross-entropy kernel
on the panel (same bytes: one read + one write of the panel, same per-element exp) - a stand-in that costs what the real pass
would. Everything else is the real thing. Tools only.
  python tools/probes/head_tiling_probe.py [out.md]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from slamkit_amd import engine as E  # noqa: E402

lib = E.load_library()
st = E.current_stream_ptr()
dev = "cuda"
M, H, V = 16384, 1536, 152167
VP = ((V + 255) // 256) * 256
p = lambda t: t.data_ptr()  # noqa: E731
lines = []


def say(x):
    print(x, flush=True)
    lines.append(x)


def timed(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator(device=dev).manual_seed(0)
hf = (torch.randn(M, H, device=dev, generator=g) * 1.0).to(torch.bfloat16)
Emb = (torch.randn(VP, H, device=dev, generator=g) * 0.02).to(torch.bfloat16)
EmbT = Emb.t().contiguous()                       # the engine's transposed weight image [H][Vp]
labels = torch.randint(0, V, (1, M), device=dev, generator=g)
row_loss = torch.empty(M, device=dev)
scr = torch.zeros(2, device=dev)
dE = torch.zeros(VP, H, device=dev, dtype=torch.float32)
dhf = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
ws = torch.empty(lib.slam_op_gemm_tn_workspace(M, VP, H) // 4 + 16, device=dev, dtype=torch.float32)

# ---- materialised -------------------------------------------------------------------------------------------------
logits = torch.empty(M, VP, device=dev, dtype=torch.bfloat16)


def mat_fwd():
    assert lib.slam_op_gemm_nt(p(hf), p(Emb), p(logits), None, None, M, VP, H, 1, st) == 0
    assert lib.slam_op_cross_entropy(p(logits), p(labels), float(M), p(logits), p(row_loss), p(scr), 1, M, VP, V, st) == 0


def mat_bwd():
    assert lib.slam_op_gemm_tn(p(logits), p(hf), p(dE), 0, M, VP, H, p(ws), st) == 0
    assert lib.slam_op_gemm_nt(p(logits), p(EmbT), p(dhf), None, None, M, H, VP, 1, st) == 0


def piece(fn):
    return timed(fn, reps=2, warm=1)


say("# LM head + cross-entropy at configs[3] (M 16384, H 1536, Vp 152320): materialised logits vs vocabulary panels\n")
t_gemm = piece(lambda: lib.slam_op_gemm_nt(p(hf), p(Emb), p(logits), None, None, M, VP, H, 1, st))
t_ce = piece(lambda: lib.slam_op_cross_entropy(p(logits), p(labels), float(M), p(logits), p(row_loss), p(scr), 1, M, VP, V, st))
t_wg = piece(lambda: lib.slam_op_gemm_tn(p(logits), p(hf), p(dE), 0, M, VP, H, p(ws), st))
t_dg = piece(lambda: lib.slam_op_gemm_nt(p(logits), p(EmbT), p(dhf), None, None, M, H, VP, 1, st))
t_mat = timed(lambda: (mat_fwd(), mat_bwd()), reps=2, warm=1)
say("| materialised pipeline | ms |\n|---|---|")
say(f"| head GEMM -> 5.0 GB logits | {t_gemm:.2f} |")
say(f"| cross-entropy in place (read + write 5.0 GB) | {t_ce:.2f} |")
say(f"| weight gradient dE = dlogits^T hf | {t_wg:.2f} |")
say(f"| dgrad dhf = dlogits E | {t_dg:.2f} |")
say(f"| **whole pipeline, back to back** | **{t_mat:.2f}** |")
del logits
torch.cuda.empty_cache()

# ---- tiled ----------------------------------------------------------------------------------------------------------
for P in (4096, 8192, 16384):
    panels = [(c0, min(P, VP - c0)) for c0 in range(0, VP, P)]
    buf = torch.empty(M, P, device=dev, dtype=torch.bfloat16)
    EmbT_p = [EmbT[:, c0:c0 + w].contiguous() for c0, w in panels]       # [H][w] panel images of the transposed weights
    lab_p = torch.randint(0, P // 2, (1, M), device=dev, generator=g)
    acc = torch.zeros(M, H, device=dev, dtype=torch.bfloat16)
    ws_p = torch.empty(max(lib.slam_op_gemm_tn_workspace(M, w, H) for _, w in panels) // 4 + 16, device=dev, dtype=torch.float32)

    def gemm_panel(i):
        c0, w = panels[i]
        assert lib.slam_op_gemm_nt(p(hf), p(Emb[c0:]), p(buf), None, None, M, w, H, 1, st) == 0

    def pass_panel(i):
        c0, w = panels[i]
        assert lib.slam_op_cross_entropy(p(buf), p(lab_p), float(M), p(buf), p(row_loss), p(scr), 1, M, w, w, st) == 0

    def tiled():
        for i in range(len(panels)):           # forward: logits panel + row statistics
            gemm_panel(i)
            pass_panel(i)
        for i, (c0, w) in enumerate(panels):   # backward: recompute, gradient pass, panel wgrad, panel dgrad (+= dhf)
            gemm_panel(i)
            pass_panel(i)
            assert lib.slam_op_gemm_tn(p(buf), p(hf), p(dE[c0:]), 0, M, w, H, p(ws_p), st) == 0
            assert lib.slam_op_gemm_nt(p(buf), p(EmbT_p[i]), p(acc), None, p(acc), M, H, w, 1, st) == 0

    t_g = piece(lambda: [gemm_panel(i) for i in range(len(panels))])
    t_p = piece(lambda: [pass_panel(i) for i in range(len(panels))])
    t_w = piece(lambda: [lib.slam_op_gemm_tn(p(buf), p(hf), p(dE[c0:]), 0, M, w, H, p(ws_p), st) for c0, w in panels])
    t_d = piece(lambda: [lib.slam_op_gemm_nt(p(buf), p(EmbT_p[i]), p(acc), None, p(acc), M, H, w, 1, st) for i, (c0, w) in enumerate(panels)])
    t_all = timed(tiled, reps=2, warm=1)
    say(f"\n| tiled pipeline, {len(panels)} panels of {P} columns ({M * P * 2 / 1e6:.0f} MB panel buffer) | ms |\n|---|---|")
    say(f"| head GEMM, all panels (runs TWICE per step) | {t_g:.2f} |")
    say(f"| element-wise pass over all panels (runs twice: statistics, gradient) | {t_p:.2f} |")
    say(f"| panel weight gradients | {t_w:.2f} |")
    say(f"| panel dgrads accumulating into dhf | {t_d:.2f} |")
    say(f"| **whole pipeline, back to back** | **{t_all:.2f}** (materialised: {t_mat:.2f}) |")
    del buf, EmbT_p, acc, ws_p
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
