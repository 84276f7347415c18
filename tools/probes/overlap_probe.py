"""Is there anything to win by running K loops beside epilogues? Two persistent 256 x 256 launches of half the rows each (M 4096:
608 tiles on 128 one-per-CU blocks = the 4.75 tiles per block of the full launch) on two streams, in phase and with the second
one started d us late, against the one full-chip launch. In phase, both halves store together (the lockstep of the full launch);
offset by half a tile period, one half's epilogues fall under the other's K loops. HIP events; random operands.
Usage: python tools/probes/overlap_probe.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slamkit_amd import engine as E
lib = E.load_library(); dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def rb(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
clock_mhz = 100.0  # torch.cuda._sleep counts device cycles of the 100 MHz wall clock on ROCm builds? calibrated below
def calibrate():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s2):
        e0.record(); torch.cuda._sleep(1000000); e1.record()
    torch.cuda.synchronize()
    return 1000000 / (e0.elapsed_time(e1) * 1e3)  # cycles per us
cpu = calibrate()
print(f"torch.cuda._sleep: {cpu:.1f} cycles per us")
for name, kind, M, N, K in [("gate|up + SwiGLU", "swiglu", 8192, 9728, 896), ("gate|up plain", "plain", 8192, 9728, 896), ("down dgrad + dSwiGLU", "dswiglu", 8192, 4864, 896)]:
    x, w = rb(M, K), rb(N, K)
    y = torch.empty(M, N if kind != "dswiglu" else 2 * N, dtype=torch.bfloat16, device=dev)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev) if kind == "swiglu" else None
    if kind == "dswiglu": y.copy_(rb(M, 2 * N))
    def launch(lo, rows, st):
        xs, ys = x[lo:lo + rows], y[lo:lo + rows]
        if kind == "plain": return lib.slam_op_gemm_nt(xs.data_ptr(), w.data_ptr(), ys.data_ptr(), None, None, rows, N, K, 1, st)
        if kind == "swiglu": return lib.slam_op_gemm_nt_swiglu(xs.data_ptr(), w.data_ptr(), ys.data_ptr(), act[lo:lo + rows].data_ptr(), rows, N, K, st)
        return lib.slam_op_gemm_nt_dswiglu(xs.data_ptr(), w.data_ptr(), ys.data_ptr(), rows, N, K, st)
    def run(mode, delay_us=0.0):
        ts = []
        for it in range(iters + 3):
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            go = torch.cuda.Event()
            with torch.cuda.stream(s1):
                torch.cuda._sleep(int(20 * cpu))  # both streams are loaded before anything starts
                e0.record(); go.record()
                if mode == "full":
                    assert launch(0, M, s1.cuda_stream) == 0
                else:
                    assert launch(0, M // 2, s1.cuda_stream) == 0
                e1.record()
            if mode != "full":
                with torch.cuda.stream(s2):
                    s2.wait_event(go)
                    if delay_us > 0: torch.cuda._sleep(int(delay_us * cpu))
                    assert launch(M // 2, M // 2, s2.cuda_stream) == 0
                    e2.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3
            if mode != "full": t = max(t, e0.elapsed_time(e2) * 1e3)
            if it >= 3: ts.append(t)
        ts.sort()
        return ts[len(ts) // 2]
    lib.slam_set_option(None, b"gemm_256_persist_cus", 0)
    full = run("full")
    lib.slam_set_option(None, b"gemm_256_persist_cus", 128)
    lib.slam_set_option(None, b"gemm_256_stagger", 0); lib.slam_set_option(None, b"gemm_256_stagger_dswiglu", 0)
    res = [(d, run("split", d)) for d in (0, 3, 6, 9, 12, 15, 20)]
    lib.slam_set_option(None, b"gemm_256_persist_cus", 0)
    lib.slam_set_option(None, b"gemm_256_stagger", 1200); lib.slam_set_option(None, b"gemm_256_stagger_dswiglu", 1200)
    print(f"{name} {M}x{N}x{K}: full chip {full:.1f} us | two 128-CU halves, second started d us late: " + "  ".join(f"[d={d}] {t:.1f}" for d, t in res), flush=True)
