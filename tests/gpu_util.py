"""Helpers for the -m gpu parity tests: call kernels through the C ABI with torch device buffers."""
import ctypes as C

import torch

from slamkit_amd import engine as E


def lib():
    return E.load_library()


def stream():
    return E.current_stream_ptr()


def ptr(t):
    return None if t is None else C.c_void_p(int(t.data_ptr()))


def dev_bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).contiguous().cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float()  # bf16-representable fp32


def sync():
    torch.cuda.synchronize()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """RMS error relative to RMS of the reference b."""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def max_err(a, b) -> float:
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def cosine(a, b) -> float:
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def check(name, got, ref, rms_tol, max_tol=None):
    r = rel_err(got, ref)
    m = max_err(got, ref)
    scale = float(ref.double().abs().max())
    print(f"[parity] {name}: rel_rms={r:.3e} max_abs={m:.3e} ref_absmax={scale:.3e}")
    assert torch.isfinite(got.float()).all(), f"{name}: non-finite values"
    assert r <= rms_tol, f"{name}: rel rms {r:.3e} > {rms_tol}"
    if max_tol is not None:
        assert m <= max_tol * max(scale, 1e-6), f"{name}: max abs {m:.3e} > {max_tol} * {scale:.3e}"
