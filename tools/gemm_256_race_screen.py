"""Race screen for the phase-scheduled kernels: many launches on fresh random data, every output compared bit for bit
with the 128 x 128 kernel (a DMA read before its wait shows up as rare wrong tiles, not as a steady error)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slamkit_amd import engine as E
lib = E.load_library(); st = E.current_stream_ptr(); dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
shapes = [(8192, 9728, 896), (8192, 4864, 896), (4096, 4096, 128), (2048, 8192, 192), (16384, 2048, 1536), (8192, 1536, 8960)]
for it in range(iters):
    M, N, K = shapes[it % len(shapes)]
    g = torch.Generator(device=dev).manual_seed(1000 + it)
    x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    outs = []
    for mode in (0, 2):
        lib.slam_set_option(None, b"gemm_256", mode)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        assert lib.slam_op_gemm_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, M, N, K, 2, st) == 0
        outs.append(y)
    torch.cuda.synchronize()
    if not torch.equal(outs[0], outs[1]):
        bad += 1
        d = (outs[0].float() - outs[1].float()).abs()
        print(f"MISMATCH it={it} shape={M}x{N}x{K} max={float(d.max())} count={int((d > 0).sum())}", flush=True)
lib.slam_set_option(None, b"gemm_256", 1)
print(f"race screen: {iters} launches per kernel, {bad} mismatches")
sys.exit(1 if bad else 0)
