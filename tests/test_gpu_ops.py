"""-m gpu parity tests, one HIP kernel at a time, each called through the C ABI and checked
against a plain fp32 PyTorch reference of the same op on the same (bf16-representable) inputs.
Tolerances: outputs are rounded to bf16 once (rel 2^-9 ~ 2e-3 rms); accumulations are fp32."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import check, dev_bf16, lib, ptr, rnd, stream, sync
from oracle import slam_oracle as O

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------ GEMM
@pytest.fixture(params=[0, 1], ids=["mfma16x16x32", "mfma32x32x16"])
def mf32(request):
    """Main-loop MFMA shape of the kernels that have both (gemm_mf32); the default is restored afterwards."""
    assert lib().slam_set_option(None, b"gemm_mf32", request.param) == 0
    yield request.param
    lib().slam_set_option(None, b"gemm_mf32", 0)


@pytest.mark.parametrize("glds", [0, 1])  # register staging / LDS-DMA
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (300, 256, 128), (1000, 1152, 896),
                                   (74, 512, 256)])
def test_gemm_nt(M, N, K, glds, mf32):
    X, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    Xd, Wd, bd, rd = dev_bf16(X), dev_bf16(W), dev_bf16(bias), dev_bf16(res)  # keep alive: ptr() borrows
    for use_bias, use_res in [(False, False), (True, True)]:
        Y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        rc = lib().slam_op_gemm_nt(ptr(Xd), ptr(Wd), ptr(Y), ptr(bd) if use_bias else None,
                                   ptr(rd) if use_res else None, M, N, K, glds, stream())
        sync()
        assert rc == 0
        ref = X @ W.t() + (bias if use_bias else 0) + (res if use_res else 0)
        check(f"gemm_nt {M}x{N}x{K} glds={glds} mf32={mf32} epi={use_bias}", Y.float(), ref, 4e-3, 2e-2)


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 128), (2048, 8192, 192), (4096, 2048, 64 * 5)])
def test_gemm_nt_256_tile_kernel(M, N, K, mf32):
    """The 256 x 256 / 8-wave / 8-phase kernel (default for the gate|up forward) against the 128 x 128 kernel: same
    bits for plain, bias + residual and fused-SwiGLU epilogues (under either MFMA shape: the contraction order of an output
    element does not depend on the tile), and both against the fp32 reference."""
    X, W, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3), rnd(M, N, seed=4)
    Xd, Wd, bd, rd = dev_bf16(X), dev_bf16(W), dev_bf16(bias), dev_bf16(res)
    outs = {}
    for mode in (0, 2):  # 0 = 128 x 128 only, 2 = 256 x 256 whenever the shape allows
        assert lib().slam_set_option(None, b"gemm_256", mode) == 0
        Y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        assert lib().slam_op_gemm_nt(ptr(Xd), ptr(Wd), ptr(Y), ptr(bd), ptr(rd), M, N, K, 2, stream()) == 0
        Yp = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        assert lib().slam_op_gemm_nt(ptr(Xd), ptr(Wd), ptr(Yp), None, None, M, N, K, 2, stream()) == 0
        Ys = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        act = torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device="cuda")
        assert lib().slam_op_gemm_nt_swiglu(ptr(Xd), ptr(Wd), ptr(Ys), ptr(act), M, N, K, stream()) == 0
        sync()
        outs[mode] = (Y, Yp, Ys, act)
    lib().slam_set_option(None, b"gemm_256", 1)
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)
    ref = X @ W.t()
    check("gemm 256 bias+resid", outs[2][0].float(), ref + bias + res, 4e-3, 2e-2)
    check("gemm 256 plain", outs[2][1].float(), ref, 4e-3, 2e-2)
    gu = ref.view(M, N // 64, 2, 32)
    check("gemm 256 swiglu act", outs[2][3].float(), (F.silu(gu[:, :, 0]) * gu[:, :, 1]).reshape(M, N // 2), 6e-3, 3e-2)


@pytest.mark.parametrize("variant", [(0, 0), (1, 0), (1, 1)], ids=["mfma16-8wave", "mfma32-8wave", "mfma32-4wave"])
@pytest.mark.parametrize("M,N,K", [(2304, 8192, 192), (4352, 4096, 64 * 5), (8192, 9728, 896), (4096, 4864, 128)])
def test_gemm_nt_256_persistent_blocks(M, N, K, variant):
    """gemm_256_persist: one block per CU walking the tile list, the DMA stream running across tile boundaries - same bits
    as one block per tile (same contraction order per tile) for the plain, fused-SwiGLU-forward and fused-SwiGLU-backward
    epilogues, on grids with ragged last rounds (288, 272, 1216, 304 tiles); the SwiGLU backward against fp32 as well.
    Round 5: for the eight-wave kernel on either MFMA shape and for the persistent four-wave kernel (128 x 128 per wave,
    32x32x16) against the one-block-per-tile eight-wave kernel of the same MFMA shape."""
    X, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    GU = rnd(M, 2 * N, seed=5)
    Xd, Wd, gud = dev_bf16(X), dev_bf16(W), dev_bf16(GU)
    outs = {}
    try:
        assert lib().slam_set_option(None, b"gemm_mf32", variant[0]) == 0
        assert lib().slam_set_option(None, b"gemm_256_w4", variant[1]) == 0
        assert lib().slam_set_option(None, b"gemm_256", 2) == 0
        for persist in (0, 1, 1):
            assert lib().slam_set_option(None, b"gemm_256_persist", persist) == 0
            Yp = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert lib().slam_op_gemm_nt(ptr(Xd), ptr(Wd), ptr(Yp), None, None, M, N, K, 2, stream()) == 0
            Ys = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            act = torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert lib().slam_op_gemm_nt_swiglu(ptr(Xd), ptr(Wd), ptr(Ys), ptr(act), M, N, K, stream()) == 0
            g = gud.clone()
            assert lib().slam_op_gemm_nt_dswiglu(ptr(Xd), ptr(Wd), ptr(g), M, N, K, stream()) == 0
            sync()
            outs.setdefault(persist, []).append((Yp, Ys, act, g))
    finally:
        lib().slam_set_option(None, b"gemm_256_persist", 1)
        lib().slam_set_option(None, b"gemm_256", 1)
        lib().slam_set_option(None, b"gemm_mf32", 0)
        lib().slam_set_option(None, b"gemm_256_w4", 0)
    for run in outs[1]:
        for a, b in zip(outs[0][0], run):
            assert torch.equal(a, b)
    d = (dev_bf16(X).float() @ dev_bf16(W).float().t()).cpu()  # d(act) [M][N]
    gu = dev_bf16(GU).float().cpu().view(M, N // 32, 2, 32)
    gate, up = gu[:, :, 0].reshape(M, N), gu[:, :, 1].reshape(M, N)
    sg = torch.sigmoid(gate)
    ref = torch.stack([(d * up * sg * (1 + gate * (1 - sg))).view(M, N // 32, 32), (d * gate * sg).view(M, N // 32, 32)], 2)
    check("gemm 256 persistent dswiglu", outs[1][0][3].float().cpu().view(M, N // 32, 2, 32), ref, 6e-3, 3e-2)


@pytest.mark.parametrize("M,N,K", [(256, 224, 128), (512, 896, 64 * 5), (1024, 448, 64 * 19), (8192, 896, 1152)])
def test_gemm_nt_224_tile_kernel(M, N, K):
    """The 256 x 224 / 8-wave / 8-phase NT kernel (dgrad launches of the two-stream backward), forced on: same bits as the
    128 x 128 kernel on the same MFMA shape (16x16x32: identical contraction order), with and without the residual epilogue;
    both against fp32."""
    X, W, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(M, N, seed=4)
    Xd, Wd, rd = dev_bf16(X), dev_bf16(W), dev_bf16(res)
    outs = {}
    try:
        assert lib().slam_set_option(None, b"gemm_mf32", 0) == 0
        for mode in (0, 2):
            assert lib().slam_set_option(None, b"gemm_nt224", mode) == 0
            assert lib().slam_set_option(None, b"gemm_256", 0) == 0
            Y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            Yr = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert lib().slam_op_gemm_nt(ptr(Xd), ptr(Wd), ptr(Y), None, None, M, N, K, 1, stream()) == 0
            assert lib().slam_op_gemm_nt(ptr(Xd), ptr(Wd), ptr(Yr), None, ptr(rd), M, N, K, 1, stream()) == 0
            sync()
            outs[mode] = (Y, Yr)
    finally:
        lib().slam_set_option(None, b"gemm_nt224", 1)
        lib().slam_set_option(None, b"gemm_256", 1)
        lib().slam_set_option(None, b"gemm_mf32", 0)
    ref = X @ W.t()
    check(f"gemm_nt_224 {M}x{N}x{K}", outs[2][0].float(), ref, 4e-3, 2e-2)
    check(f"gemm_nt_224 resid {M}x{N}x{K}", outs[2][1].float(), ref + res, 4e-3, 2e-2)
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])


def test_gemm_nt_operand_beyond_4GB():
    """A row operand of more than 4 GB (the d-logits of the configs[3] micro-batch: [16384][152320] bf16 = 4.99 GB): the
    per-lane 32-bit DMA offsets are relative to the tile, the 64-bit part rides in the wave-uniform base. Regression for a
    round-1 bug (offsets from the matrix origin wrapped for rows >= 14,099; found by the segment-permutation property of
    test_configs3_full_depth_packed_vs_oracle). Both NT kernels; the last and the first rows against an fp32 reference."""
    M, N, K = 16384, 1024, 152320
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    for r in range(0, M, 2048):
        X[r:r + 2048] = (torch.randn(2048, K, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    assert X.numel() * 2 > 2 ** 32
    Wf = W.float().cpu()
    try:
        for mode in (0, 2):  # 128 x 128 kernel, 256 x 256 kernel
            assert lib().slam_set_option(None, b"gemm_256", mode) == 0
            Y = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert lib().slam_op_gemm_nt(ptr(X), ptr(W), ptr(Y), None, None, M, N, K, 1, stream()) == 0
            sync()
            for lo in (0, 14080, M - 128):
                ref = X[lo:lo + 128].float().cpu() @ Wf.t()
                check(f"gemm_nt >4GB rows {lo}.. gemm_256={mode}", Y[lo:lo + 128].float(), ref, 4e-3, 2e-2)
    finally:
        lib().slam_set_option(None, b"gemm_256", 1)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (256, 192, 384), (300, 512, 256), (1000, 1152, 896),
                                   (74, 1024, 256)])
def test_gemm_nn(M, N, K):
    dY, W, res = rnd(M, N, seed=5), rnd(N, K, seed=6, scale=0.05), rnd(M, K, seed=7)
    dYd, Wd, rd = dev_bf16(dY), dev_bf16(W), dev_bf16(res)
    for use_res in (False, True):
        dX = torch.full((M, K), float("nan"), dtype=torch.bfloat16, device="cuda")
        rc = lib().slam_op_gemm_nn(ptr(dYd), ptr(Wd), ptr(dX), ptr(rd) if use_res else None, M, N, K, stream())
        sync()
        assert rc == 0
        check(f"gemm_nn {M}x{N}x{K} res={use_res}", dX.float(), dY @ W + (res if use_res else 0), 4e-3, 2e-2)


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (256, 128, 256), (300, 384, 128), (2048, 1152, 896),
                                   (74, 512, 256), (1111, 128, 128)])
def test_gemm_tn(M, N, K):
    dY, X = rnd(M, N, seed=8), rnd(M, K, seed=9)
    ws = torch.empty(lib().slam_op_gemm_tn_workspace(M, N, K) // 4 + 16, dtype=torch.float32, device="cuda")
    dW = torch.full((N, K), 1.0, dtype=torch.float32, device="cuda")
    a, b = dev_bf16(dY), dev_bf16(X)
    assert lib().slam_op_gemm_tn(ptr(a), ptr(b), ptr(dW), 0, M, N, K, ptr(ws), stream()) == 0
    sync()
    ref = dY.t() @ X
    check(f"gemm_tn {M}x{N}x{K}", dW, ref, 1e-5, 1e-4)
    assert lib().slam_op_gemm_tn(ptr(a), ptr(b), ptr(dW), 1, M, N, K, ptr(ws), stream()) == 0
    sync()
    check(f"gemm_tn accumulate {M}x{N}x{K}", dW, 2 * ref, 1e-5, 1e-4)


@pytest.mark.parametrize("M,N,K", [(64, 256, 224), (320, 512, 448), (2048, 768, 672), (1024, 224, 512), (4096, 448, 1024),
                                   (8192, 9728, 896), (8192, 896, 4864)])
def test_gemm_tn_224_phase_scheduled(M, N, K):
    """256 x 224 / 8-wave / 8-phase wgrad kernel (both store orientations, forced on: every split plan from one piece to
    16 pieces per tile occurs across these shapes) against the fp32 reference and - bit for bit across repeats - itself."""
    dY, X = rnd(M, N, seed=8), rnd(M, K, seed=9)
    a, b = dev_bf16(dY), dev_bf16(X)
    ref = dY.t() @ X
    try:
        assert lib().slam_set_option(None, b"gemm_tn224", 2) == 0
        ws = torch.empty(lib().slam_op_gemm_tn_workspace(M, N, K) // 4 + 16, dtype=torch.float32, device="cuda")
        outs = []
        for rep in range(3):
            ws.fill_(float("nan"))
            dW = torch.full((N, K), 1.0, dtype=torch.float32, device="cuda")
            assert lib().slam_op_gemm_tn(ptr(a), ptr(b), ptr(dW), 0, M, N, K, ptr(ws), stream()) == 0
            sync()
            outs.append(dW.clone())
            check(f"gemm_tn_224 {M}x{N}x{K}", dW, ref, 1e-5, 1e-4)
            assert lib().slam_op_gemm_tn(ptr(a), ptr(b), ptr(dW), 1, M, N, K, ptr(ws), stream()) == 0
            sync()
            check(f"gemm_tn_224 accumulate {M}x{N}x{K}", dW, 2 * ref, 1e-5, 1e-4)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    finally:
        lib().slam_set_option(None, b"gemm_tn224", 1)


@pytest.mark.parametrize("M,N,K", [(256, 512, 256), (1600, 512, 256), (8192, 1152, 896), (8192, 896, 896), (8192, 512, 896),   # balanced 128 x 128 plans
                                   (4096, 1024, 896), (8192, 9728, 896), (8192, 896, 4864), (2048, 768, 672),                # 256 x 224, both orientations
                                   (333, 72, 40)])                                                                            # generic split-K fallback
@pytest.mark.parametrize("background", [0, 1])
def test_gemm_tn_bf16_image_is_the_rounded_result(M, N, K, background):
    """slam_set_grad_image's mechanism, kernel by kernel: whichever kernel stores the FINAL fp32 value of a dW element (an
    unsplit tile's epilogue, or the slab reduce of a split tile - every plan occurs across these shapes, in the foreground
    and in the side stream's "background" planning) also stores its bf16 rounding: the image equals dW.to(bf16) bit for bit,
    on the storing pass and on an accumulating one."""
    dY, X = rnd(M, N, seed=8), rnd(M, K, seed=9)
    a, b = dev_bf16(dY), dev_bf16(X)
    ws = torch.empty(lib().slam_op_gemm_tn_workspace(M, N, K) // 4 + 16, dtype=torch.float32, device="cuda")
    dW = torch.full((N, K), 3.0, dtype=torch.float32, device="cuda")
    try:
        for tn224 in (1, 2):
            assert lib().slam_set_option(None, b"gemm_tn224", tn224) == 0
            for acc in (0, 1):
                img = torch.full((N, K), 7.0, dtype=torch.bfloat16, device="cuda")
                assert lib().slam_op_gemm_tn_image(ptr(a), ptr(b), ptr(dW), ptr(img), acc, M, N, K, ptr(ws), background, stream()) == 0
                sync()
                assert torch.equal(img, dW.to(torch.bfloat16)), (tn224, acc, float((img.float() - dW).abs().max()))
        check(f"gemm_tn image run {M}x{N}x{K}", dW, 2 * (dY.t() @ X), 1e-5, 1e-4)
    finally:
        lib().slam_set_option(None, b"gemm_tn224", 1)


# --------------------------------------------------------------------------------------- RMSNorm
@pytest.mark.parametrize("M,H", [(5, 256), (300, 896), (1000, 1536), (70, 512), (4099, 2048), (8192, 896), (9001, 1536), (16384, 1536)])
def test_rmsnorm_fwd_bwd(M, H):
    x, w, dy, dres = rnd(M, H, seed=1), 1 + 0.1 * rnd(H, seed=2), rnd(M, H, seed=3), rnd(M, H, seed=4)
    w = w.to(torch.bfloat16).float()
    xd, wd = dev_bf16(x), dev_bf16(w)
    y = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
    rstd = torch.empty(M, dtype=torch.float32, device="cuda")
    assert lib().slam_op_rmsnorm_fwd(ptr(xd), ptr(wd), ptr(y), ptr(rstd), M, H, 1e-6, stream()) == 0
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = O.rms_norm(xr, wr, 1e-6)
    sync()
    check("rmsnorm_fwd", y.float(), yr.detach(), 3e-3, 1e-2)
    check("rmsnorm rstd", rstd, torch.rsqrt(x.pow(2).mean(-1) + 1e-6), 1e-6)
    yr.backward(dy)
    ws = torch.empty(lib().slam_op_rmsnorm_bwd_workspace(M, H) // 4 + 16, dtype=torch.float32, device="cuda")
    dx = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
    dw = torch.full((H,), 7.0, dtype=torch.float32, device="cuda")
    dyd, dresd = dev_bf16(dy), dev_bf16(dres)
    for use_res in (False, True):
        assert lib().slam_op_rmsnorm_bwd(ptr(dyd), ptr(xd), ptr(wd), ptr(rstd), ptr(dresd) if use_res else None,
                                         ptr(dx), ptr(dw), ptr(ws), M, H, stream()) == 0
        sync()
        check(f"rmsnorm_bwd dx res={use_res}", dx.float(), xr.grad + (dres if use_res else 0), 3e-3, 1e-2)
        check("rmsnorm_bwd dw", dw, wr.grad, 1e-5)


# ------------------------------------------------------------------------------------------ RoPE
@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("packed", [False, True])
def test_rope(packed, hd):
    B, T, nH, nKV = 2, 75, 4, 2
    M, ld = B * T, (nH + 2 * nKV) * hd
    qkv = rnd(M, ld, seed=1)
    if packed:
        pos = torch.cat([torch.arange(40), torch.arange(60), torch.arange(50)])[None]
    else:
        pos = torch.arange(T)[None].expand(B, T)
    buf = dev_bf16(qkv)
    cs = torch.empty(2 * M * (hd // 2), dtype=torch.float32, device="cuda")
    posd = pos.reshape(-1).contiguous().cuda() if packed else None
    assert lib().slam_op_rope(ptr(buf), ld, M, T, nH + nKV, hd, ptr(posd), 10000.0, 0, ptr(cs), stream()) == 0
    sync()
    q = qkv[:, : nH * hd].view(1, M, nH, hd).transpose(1, 2)
    k = qkv[:, nH * hd: (nH + nKV) * hd].view(1, M, nKV, hd).transpose(1, 2)
    cos, sin = O.rope_cos_sin(pos.reshape(1, M), hd, 10000.0)
    qr, kr = O.apply_rope(q, k, cos, sin)
    ref = qkv.clone()
    ref[:, : nH * hd] = qr.transpose(1, 2).reshape(M, nH * hd)
    ref[:, nH * hd: (nH + nKV) * hd] = kr.transpose(1, 2).reshape(M, nKV * hd)
    check("rope fwd", buf.float(), ref, 3e-3, 1e-2)
    # backward = transpose rotation: applying it to the forward result returns the input
    assert lib().slam_op_rope(ptr(buf), ld, M, T, nH + nKV, hd, ptr(posd), 10000.0, 1, ptr(cs), stream()) == 0
    sync()
    check("rope bwd(fwd(x)) == x", buf.float(), qkv, 6e-3, 3e-2)


@pytest.mark.parametrize("M,N,K", [(2304, 8192, 192), (4352, 4096, 64 * 5), (8192, 9728, 896), (8192, 4864, 896), (4096, 4864, 128)])
def test_gemm_nt_256_loader_storer_roles(M, N, K):
    """gemm_256_roles (round 6): the persistent 256 x 256 kernel with one wave row issuing every LDS-DMA and handing its results
    over through LDS, the other storing everything - the same tiles, the same contraction order, the same epilogue arithmetic:
    plain, fused-SwiGLU-forward and fused-SwiGLU-backward outputs must equal the undivided persistent kernel's BIT FOR BIT
    (twice: the staging region is reused across tiles), on grids with ragged last rounds and at the step's own shapes."""
    X, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    GU = rnd(M, 2 * N, seed=5)
    Xd, Wd, gud = dev_bf16(X), dev_bf16(W), dev_bf16(GU)
    outs = {}
    try:
        assert lib().slam_set_option(None, b"gemm_256", 2) == 0
        for roles in (0, 1, 1):
            assert lib().slam_set_option(None, b"gemm_256_roles", roles) == 0
            Yp = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert lib().slam_op_gemm_nt(ptr(Xd), ptr(Wd), ptr(Yp), None, None, M, N, K, 2, stream()) == 0
            Ys = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            act = torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert lib().slam_op_gemm_nt_swiglu(ptr(Xd), ptr(Wd), ptr(Ys), ptr(act), M, N, K, stream()) == 0
            g = gud.clone()
            assert lib().slam_op_gemm_nt_dswiglu(ptr(Xd), ptr(Wd), ptr(g), M, N, K, stream()) == 0
            sync()
            outs.setdefault(roles, []).append((Yp, Ys, act, g))
    finally:
        lib().slam_set_option(None, b"gemm_256_roles", 0)
        lib().slam_set_option(None, b"gemm_256", 1)
    for run in outs[1]:
        for name, a, b in zip(("plain", "swiglu C", "swiglu act", "dswiglu"), outs[0][0], run):
            assert torch.equal(a, b), f"{name}: {int((a != b).sum())} of {a.numel()} elements differ"


# ---------------------------------------------------------------------------------------- SwiGLU
def test_swiglu():
    M, I = 130, 512
    gu, dact = rnd(M, 2 * I, seed=1, scale=2.0), rnd(M, I, seed=2)
    gud = dev_bf16(gu)
    act = torch.empty(M, I, dtype=torch.bfloat16, device="cuda")
    assert lib().slam_op_swiglu_fwd(ptr(gud), ptr(act), M, I, stream()) == 0
    g = gu[:, :I].clone().requires_grad_(True)
    u = gu[:, I:].clone().requires_grad_(True)
    ref = F.silu(g) * u
    ref.backward(dact)
    sync()
    check("swiglu fwd", act.float(), ref.detach(), 3e-3, 1e-2)
    dactd = dev_bf16(dact)
    assert lib().slam_op_swiglu_bwd(ptr(gud), ptr(dactd), M, I, stream()) == 0
    sync()
    check("swiglu bwd", gud.float(), torch.cat([g.grad, u.grad], 1), 3e-3, 1e-2)


# ------------------------------------------------------------------------------------- attention
def _attn_case(seg_lens, nH, nKV, seed=0, spike=False, hd=64):
    M = sum(seg_lens)
    ld = (nH + 2 * nKV) * hd
    qkv = rnd(M, ld, seed=seed)
    if spike:  # force big running-max jumps in the online softmax
        qkv[5, :hd] *= 8.0   # powers of two keep the values bf16-representable
        qkv[3, nH * hd: nH * hd + hd] *= 8.0
    starts = []
    s = 0
    for n in seg_lens:
        starts += [s] * n
        s += n
    seg_s = torch.tensor(starts, dtype=torch.int32)
    ends = []
    s = 0
    for n in seg_lens:
        ends += [s + n] * n
        s += n
    seg_e = torch.tensor(ends, dtype=torch.int32)
    return M, ld, qkv, seg_s, seg_e


def _attn_prescale(qkv, nH, hd):
    """The attention ops take the QUERY columns pre-scaled by head_dim^-0.5 * log2(e) (the QKV projection's RoPE epilogue does
    that in the engine, one rounding). Returns (device input with bf16 q_hat = bf16(q * c2), fp32 reference input whose query
    columns are exactly q_hat / c2 - what the kernels effectively see)."""
    c2 = hd ** -0.5 * 1.4426950408889634
    dev = qkv.clone()
    dev[:, : nH * hd] = (qkv[:, : nH * hd] * c2).to(torch.bfloat16).float()
    ref = qkv.clone()
    ref[:, : nH * hd] = dev[:, : nH * hd] / c2
    return dev, ref


def _attn_ref(qkv, seg_s, nH, nKV, d_o=None, hd=64):
    M = qkv.shape[0]
    x = qkv.clone().requires_grad_(True)
    q = x[:, : nH * hd].view(1, M, nH, hd).transpose(1, 2)
    k = x[:, nH * hd: (nH + nKV) * hd].view(1, M, nKV, hd).transpose(1, 2)
    v = x[:, (nH + nKV) * hd:].view(1, M, nKV, hd).transpose(1, 2)
    i = torch.arange(M)
    mask = ((i[None, :] <= i[:, None]) & (i[None, :] >= seg_s.long()[:, None]))[None]
    o = O.attention(q, k, v, mask, hd ** -0.5).reshape(M, nH * hd)
    if d_o is None:
        return o.detach(), None
    o.backward(d_o)
    return o.detach(), x.grad


@pytest.mark.parametrize("seg_lens,nH,nKV,spike,hd", [
    ([64], 2, 1, False, 64), ([128], 2, 2, False, 64), ([200], 4, 2, True, 64), ([256, 256], 14, 2, False, 64),
    ([37, 100, 5, 130, 64], 4, 2, False, 64), ([1024], 7, 1, False, 64), ([29, 41, 17], 4, 2, True, 64),
    ([64], 2, 1, False, 128), ([200], 4, 2, True, 128), ([256, 256], 12, 2, False, 128),
    ([37, 100, 5, 130, 64], 4, 2, False, 128), ([1024], 6, 1, False, 128), ([29, 41, 17], 4, 2, True, 128)])
def test_attention_fwd_bwd(seg_lens, nH, nKV, spike, hd):
    M, ld, qkv, seg_s, seg_e = _attn_case(seg_lens, nH, nKV, seed=len(seg_lens), spike=spike, hd=hd)
    d_o = rnd(M, nH * hd, seed=9)
    qkv_dev, qkv = _attn_prescale(qkv, nH, hd)
    o_ref, dqkv_ref = _attn_ref(qkv, seg_s, nH, nKV, d_o, hd=hd)
    qd = dev_bf16(qkv_dev)
    o = torch.full((M, nH * hd), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(nH * M, dtype=torch.float32, device="cuda")
    ss, se = seg_s.cuda(), seg_e.cuda()
    assert lib().slam_op_attn_fwd(ptr(qd), ptr(o), ptr(lse), ptr(ss), M, nH, nKV, hd, stream()) == 0
    sync()
    check(f"attn fwd {seg_lens} nH={nH} hd={hd}", o.float(), o_ref, 6e-3, 3e-2)
    # lse check (log2 domain) against the scaled scores
    q = qkv[:, : nH * hd].view(M, nH, hd).transpose(0, 1)
    k = qkv[:, nH * hd: (nH + nKV) * hd].view(M, nKV, hd).transpose(0, 1).repeat_interleave(nH // nKV, 0)
    s = (q @ k.transpose(1, 2)) * hd ** -0.5
    i = torch.arange(M)
    mask = (i[None, :] <= i[:, None]) & (i[None, :] >= seg_s.long()[:, None])
    s = s.masked_fill(~mask[None], float("-inf"))
    check("attn lse2", lse.view(nH, M).cpu(), torch.logsumexp(s, -1) * 1.4426950408889634, 1e-4)
    # backward uses the bf16 O the forward produced (as the engine does)
    ws = torch.empty(lib().slam_op_attn_bwd_workspace(M, nH, hd) // 4 + 16, dtype=torch.float32, device="cuda")
    dqkv = torch.full((M, ld), float("nan"), dtype=torch.bfloat16, device="cuda")
    dod = dev_bf16(d_o)
    assert lib().slam_op_attn_bwd(ptr(qd), ptr(o), ptr(dod), ptr(lse), ptr(dqkv), ptr(ws), ptr(ss), ptr(se),
                                  M, nH, nKV, hd, stream()) == 0
    sync()
    got = dqkv.float().cpu()
    check("attn bwd dq", got[:, : nH * hd], dqkv_ref[:, : nH * hd], 1.5e-2, 6e-2)
    check("attn bwd dk", got[:, nH * hd: (nH + nKV) * hd], dqkv_ref[:, nH * hd: (nH + nKV) * hd], 1.5e-2, 6e-2)
    check("attn bwd dv", got[:, (nH + nKV) * hd:], dqkv_ref[:, (nH + nKV) * hd:], 1.5e-2, 6e-2)


_ATTN_REF_CACHE = {}


@pytest.mark.parametrize("jq,kw,nch", [(2, 1, 4), (2, 1, 1), (1, 1, 1), (1, 1, 2), (2, 1, 3)])
@pytest.mark.parametrize("seg_lens,nH,nKV", [([256, 256], 14, 2), ([37, 100, 5, 130, 64], 4, 2), ([1024], 7, 1),
                                             ([29, 41, 17], 4, 2), ([700, 324], 6, 3)])
def test_attention_bwd_launch_shapes(seg_lens, nH, nKV, jq, kw, nch):
    """Every launch shape of the backward (rows per wave in dQ, keys per wave in dK/dV, query-range chunks per key tile)
    against the fp32 reference, and bit-reproducible; head_dim 64 (the shapes only exist there)."""
    hd = 64
    M, ld, qkv, seg_s, seg_e = _attn_case(seg_lens, nH, nKV, seed=len(seg_lens), hd=hd)
    d_o = rnd(M, nH * hd, seed=9)
    qkv_dev, qkv = _attn_prescale(qkv, nH, hd)
    key = (tuple(seg_lens), nH, nKV)
    if key not in _ATTN_REF_CACHE:
        _ATTN_REF_CACHE[key] = _attn_ref(qkv, seg_s, nH, nKV, d_o, hd=hd)
    o_ref, dqkv_ref = _ATTN_REF_CACHE[key]
    qd, dod = dev_bf16(qkv_dev), dev_bf16(d_o)
    o = torch.empty(M, nH * hd, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(nH * M, dtype=torch.float32, device="cuda")
    ss, se = seg_s.cuda(), seg_e.cuda()
    L = lib()
    try:
        for k, v in (("attn_jq", jq), ("attn_kw", kw), ("attn_nch", nch)):
            assert L.slam_set_option(None, k.encode(), v) == 0
        assert L.slam_op_attn_fwd(ptr(qd), ptr(o), ptr(lse), ptr(ss), M, nH, nKV, hd, stream()) == 0
        ws = torch.empty(L.slam_op_attn_bwd_workspace(M, nH, hd) // 4 + 16, dtype=torch.float32, device="cuda")
        outs = []
        for _ in range(2):
            dqkv = torch.full((M, ld), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert L.slam_op_attn_bwd(ptr(qd), ptr(o), ptr(dod), ptr(lse), ptr(dqkv), ptr(ws), ptr(ss), ptr(se),
                                      M, nH, nKV, hd, stream()) == 0
            sync()
            outs.append(dqkv)
    finally:
        for k, v in (("attn_jq", 1), ("attn_kw", 1), ("attn_nch", 4)):
            L.slam_set_option(None, k.encode(), v)
    assert torch.equal(outs[0], outs[1]), "attention backward is not bit-reproducible"
    got = outs[0].float().cpu()
    check("attn bwd dq", got[:, : nH * hd], dqkv_ref[:, : nH * hd], 1.5e-2, 6e-2)
    check("attn bwd dk", got[:, nH * hd: (nH + nKV) * hd], dqkv_ref[:, nH * hd: (nH + nKV) * hd], 1.5e-2, 6e-2)
    check("attn bwd dv", got[:, (nH + nKV) * hd:], dqkv_ref[:, (nH + nKV) * hd:], 1.5e-2, 6e-2)


# --------------------------------------------------------------------------------- cross entropy
@pytest.mark.parametrize("V,Vp", [(502, 512), (700, 768), (5003, 5120)])
@pytest.mark.parametrize("num_items", [0.0, 57.0])
def test_cross_entropy(num_items, V, Vp):
    B, T = 3, 41
    M = B * T
    logits = rnd(M, Vp, seed=1, scale=3.0)
    labels = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(2))
    labels[0, 5:9] = -100
    labels[2, 30:] = -100
    lg = dev_bf16(logits)
    dl = torch.full((M, Vp), float("nan"), dtype=torch.bfloat16, device="cuda")
    rl = torch.empty(M, dtype=torch.float32, device="cuda")
    sc = torch.zeros(2, dtype=torch.float32, device="cuda")
    labd = labels.cuda()
    assert lib().slam_op_cross_entropy(ptr(lg), ptr(labd), num_items, ptr(dl), ptr(rl), ptr(sc), B, T, Vp, V,
                                       stream()) == 0
    sync()
    x = logits[:, :V].view(B, T, V).clone().requires_grad_(True)
    loss = O.compute_loss(x, labels, num_items_in_batch=(num_items if num_items > 0 else None))
    loss.backward()
    assert abs(float(sc[1]) - float(loss)) <= 2e-5 * max(1.0, abs(float(loss)))
    got = dl.float().cpu().view(B, T, Vp)
    check("ce dlogits", got[:, :, :V], x.grad, 5e-3, 2e-2)
    assert float(got[:, :, V:].abs().max()) == 0.0


# ------------------------------------------------- gather-side embedding gradient, large vocabulary
@pytest.mark.parametrize("M,V,Vp,H,hot", [(300, 700, 768, 64, False), (1000, 5003, 5120, 256, True), (777, 1000, 1024, 1536, True)])
def test_embed_bwd_scatter(M, V, Vp, H, hot):
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (M,), generator=g)
    if hot:  # a few very frequent ids and the suppressed padding id
        ids[::3] = 7
        ids[1::7] = 0
        ids[5::11] = V - 1
    dh = rnd(M, H, seed=3)
    dE0 = torch.randn(Vp, H, generator=g)
    ref = dE0.clone()
    sel = ids != 0
    ref.index_add_(0, ids[sel], dh[sel])
    dhd, idd = dev_bf16(dh), ids.cuda()
    dE = dE0.clone().cuda()
    ws = torch.empty(lib().slam_op_embed_bwd_workspace(M, Vp) + 64, dtype=torch.uint8, device="cuda")
    assert lib().slam_op_embed_bwd(ptr(idd), ptr(dhd), ptr(dE), M, H, Vp, V, 0, ptr(ws), stream()) == 0
    sync()
    check("embed bwd scatter", dE.cpu(), ref, 2e-5, 2e-4)
    # bit-identical on a second run (token-ordered sums, no float atomics)
    dE2 = dE0.clone().cuda()
    assert lib().slam_op_embed_bwd(ptr(idd), ptr(dhd), ptr(dE2), M, H, Vp, V, 0, ptr(ws), stream()) == 0
    sync()
    assert torch.equal(dE, dE2)
