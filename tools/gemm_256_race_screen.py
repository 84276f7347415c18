"""Race screen for the phase-scheduled kernels: many launches on fresh random data (a DMA read before its wait shows up as
rare wrong tiles, not as a steady error).
  gemm_nt_256_kernel (one block per tile and persistent blocks), gemm_nt_224_kernel : every output bit for bit against the
  128 x 128 kernel (same contraction order);
  gemm_tn_224_kernel (both orientations, with and without K-splitting): against the balanced 128 x 128 wgrad kernel to fp32
  summation-order tolerance per element, and bit for bit against a second launch of itself.
Usage: python tools/gemm_256_race_screen.py [launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr(); dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
nt_shapes = [(8192, 9728, 896, "gemm_256"), (8192, 4864, 896, "gemm_256"), (4096, 4096, 128, "gemm_256"), (2048, 8192, 192, "gemm_256"),
             (16384, 2048, 1536, "gemm_256"), (8192, 1536, 8960, "gemm_256"),
             (8192, 896, 9728, "gemm_nt224"), (8192, 896, 1152, "gemm_nt224"), (4096, 448, 4864, "gemm_nt224"), (1024, 1792, 256, "gemm_nt224")]
for it in range(iters):
    M, N, K, opt = nt_shapes[it % len(nt_shapes)]
    g = torch.Generator(device=dev).manual_seed(1000 + it)
    x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    # (MFMA shape, tile kernel, persistent, four-wave): every output bit for bit against the 128 x 128 kernel on the SAME MFMA shape
    # (the 256 x 224 kernel has the 16x16x32 main loop only)
    for mf in ((0, 1) if opt == "gemm_256" else (0,)):
        lib.slam_set_option(None, b"gemm_mf32", mf)
        outs = []
        for mode, persist, w4 in ((0, 0, 0), (2, 0, 0), (2, 1, 0)) + (((2, 1, 1),) if mf else ()):
            lib.slam_set_option(None, b"gemm_256", mode if opt == "gemm_256" else 0)
            lib.slam_set_option(None, b"gemm_256_persist", persist)
            lib.slam_set_option(None, b"gemm_256_w4", w4)
            lib.slam_set_option(None, b"gemm_nt224", mode if opt == "gemm_nt224" else 0)
            y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            assert lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 2, st) == 0
            outs.append(y)
        torch.cuda.synchronize()
        for k, o in enumerate(outs[1:]):
            if not torch.equal(outs[0], o):
                bad += 1
                d = (outs[0].float() - o.float()).abs()
                print(f"MISMATCH nt it={it} {opt} mf32={mf} variant={k} shape={M}x{N}x{K} max={float(d.max())} count={int((d > 0).sum())}", flush=True)
    lib.slam_set_option(None, b"gemm_mf32", 0); lib.slam_set_option(None, b"gemm_256_w4", 0)
lib.slam_set_option(None, b"gemm_256", 1); lib.slam_set_option(None, b"gemm_nt224", 1); lib.slam_set_option(None, b"gemm_256_persist", 1)
tn_shapes = [(8192, 9728, 896), (8192, 896, 4864), (4096, 512, 448), (16384, 1536, 8960), (2048, 1792, 1024)]
for it in range(iters):
    M, N, K = tn_shapes[it % len(tn_shapes)]
    g = torch.Generator(device=dev).manual_seed(5000 + it)
    dy = (torch.randn(M, N, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    lib.slam_set_option(None, b"gemm_tn224", 2)
    ws = torch.empty(lib.slam_op_gemm_tn_workspace(M, N, K) // 4 + 16, dtype=torch.float32, device=dev)
    res = {}
    for name, tn224, ms in (("bal", 0, 16), ("t224", 2, 16), ("t224b", 2, 16), ("t224s1", 2, 1)):
        lib.slam_set_option(None, b"gemm_tn224", tn224); lib.slam_set_option(None, b"gemm_tn224_max_split", ms)
        dw = torch.full((N, K), float("nan"), dtype=torch.float32, device=dev)
        assert lib.slam_op_gemm_tn(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), 0, M, N, K, ws.data_ptr(), st) == 0
        res[name] = dw
    torch.cuda.synchronize()
    scale = float(res["bal"].abs().max())
    for name in ("t224", "t224s1"):
        d = float((res[name] - res["bal"]).abs().max())
        if not (d <= 2e-5 * scale):
            bad += 1
            print(f"MISMATCH tn it={it} {name} shape={M}x{N}x{K} max|d|={d} scale={scale}", flush=True)
    if not torch.equal(res["t224"], res["t224b"]):
        bad += 1
        print(f"NOT REPRODUCIBLE tn it={it} shape={M}x{N}x{K}", flush=True)
lib.slam_set_option(None, b"gemm_tn224", 1); lib.slam_set_option(None, b"gemm_tn224_max_split", 16)
print(f"race screen: {iters} launches per kernel family, {bad} mismatches")
sys.exit(1 if bad else 0)
