#!/usr/bin/env python
"""bench.py - train tokens/sec of the Slam-358M pre-training step on N MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 50 --warmup 10      (the defaults: SURVEY.md §8d protocol)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one full optimizer step of configs[1] (Slam-358M, unit_hubert_25 vocab 502, ctx 1024, bf16,
per-GPU micro-batch 8, GA 1): forward + shifted CE + backward + (N>1: bucketed RCCL gradient all-reduce
overlapped with backward) + global-norm clip 0.5 + AdamW (bf16 parameters and moments: the recipe's precision), on synthetic unit-token batches already
resident in HBM. Weak scaling (per-GPU work fixed). Prints ONE JSON line on rank 0.

Extra objects on the line:
  roofline     - dominant kernel (the gate|up projection GEMM with fused SwiGLU, M=8192 N=9728 K=896, 256x256 8-phase kernel): algorithmic flops per
                 launch / mean launch time measured here with HIP events on the launch stream, against the
                 2.5 PFLOP/s dense bf16 MFMA peak (MI355X_MICROARCH.md). `step_frac` is the whole-step
                 figure: tokens/s x 2.282 GFLOP/token / peak.
                 `roofline.in_step` - every launch family of the step as it runs IN the step (HIP timing events around each
                 launch on its own stream, slam_family_ms), dominant family by time named; `roofline.kernels` - the same
                 families as stand-alone launches.
  hbm_kernels  - the memory-bound kernels (RMSNorm forward / backward, AdamW, gradient norm) timed here: algorithmic bytes /
                 time against the 8 TB/s HBM peak; `extras` - GA = 16, the host boundary, and the fp32-master optimizer (the
                 headline runs the recipe's own bf16 optimizer state), measured after the timed region;
                 config.ms_per_step_median - median of the per-step HIP-event times.
                 Round 5: `roofline.dominant_by_time` names the kernel with the most launch time per step beside the gate|up
                 entry; `roofline.traffic` is a RECORDED counter figure (profile and commit in `traffic_source`);
                 `config.ms_per_step_median_50` = median of 50 further steps after the timed region (SURVEY.md §8d protocol);
                 `value_fp32_master_optimizer` = the step of rounds 1-3 (fp32 master + moments), like for like;
                 N > 1: `extras.dp_variants` (all_reduce / rs_ag, bf16 / fp32 wire, 2 / 4 / 8 layers per bucket: median step and
                 exposed communication time of each, measured after the timed region, under a watchdog) + `extras.nccl_env`.
  cpu_baseline - the fp32 CPU oracle (oracle/slam_oracle.py, a port of the reference step: it cannot run the
                 reference's cli/train.py itself) timed on this box's host cores on a bounded
                 sample (B=1, T=1024 fwd+bwd+clip+AdamW, median of 3 steps after a warm-up: 10-15 s of CPU work).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises: see slamkit_amd/__init__.py
os.environ.setdefault("SLAM_ALLOW_FEW_HW_QUEUES", "1")  # a launcher that pinned fewer queues gets a warning and a slower run, not a lost measurement

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TOKEN = 2.282e9       # fwd+bwd, causal-exact, T=1024
PEAK_BF16 = 2.5e15             # dense MFMA peak, MI355X_MICROARCH.md
B, T, V = 8, 1024, 502


def synth_batch(rank: int, i: int, device):
    g = torch.Generator().manual_seed(1234 + rank + 1000 * i)
    ids = torch.randint(2, V, (B, T), generator=g)
    ids[:, 0] = 1
    ids = ids.to(device)
    return {"input_ids": ids, "labels": ids}


def synth_packed_358m(rank: int, i: int, device):
    """The recipe's own data mode at Slam-358M (/root/reference README.md:89 `data.packing=true`, slamkit/data/hf_dataset.py:61-62
    DataCollatorWithFlattening; SURVEY.md §8d config 2 packed variant): ONE flattened row [1, 8192] per micro-batch, segment
    lengths ~U{64..1024} (the last one cut to fit), position_ids restarting per segment, labels -100 at segment starts."""
    g = torch.Generator().manual_seed(1234 + rank + 1000 * i)
    total, ids, pos, lab, lens = B * T, [], [], [], []
    while total > 0:
        n = min(total, int(torch.randint(64, T + 1, (1,), generator=g)))
        t = torch.randint(2, V, (n,), generator=g)
        t[0] = 1
        l = t.clone()
        l[0] = -100
        ids.append(t); pos.append(torch.arange(n)); lab.append(l); lens.append(n)
        total -= n
    cat = lambda xs: torch.cat(xs)[None].to(device)  # noqa: E731
    return {"input_ids": cat(ids), "position_ids": cat(pos), "labels": cat(lab)}, lens


def synth_padded_358m(rank: int, i: int, device):
    """SURVEY.md §8d config 2 padded variant (DataCollatorForLanguageModeling, hf_dataset.py:63-64): [8, 1024] rows of lengths
    ~U{256..1024}, right-padded with id 0, labels -100 on the padding, seed 4321."""
    g = torch.Generator().manual_seed(4321 + rank + 1000 * i)
    ids = torch.zeros(B, T, dtype=torch.long)
    lab = torch.full((B, T), -100, dtype=torch.long)
    lens = []
    for b in range(B):
        n = int(torch.randint(256, T + 1, (1,), generator=g))
        t = torch.randint(2, V, (n,), generator=g)
        t[0] = 1
        ids[b, :n] = t
        lab[b, :n] = t
        lens.append(n)
    return {"input_ids": ids.to(device), "labels": lab.to(device)}, lens


# --workload qwen1p5b (BASELINE.json configs[3], not the headline metric): Qwen2.5-1.5B-shaped body, mixed
# unit + BPE vocabulary of 152,167 ids, ctx 2048, packed sequences (flattening collator layout)
W4 = dict(name="Qwen/Qwen2.5-1.5B", vocab=152167, ctx=2048, tokens=16384, unit_lo=151667, n_mm=None)


def synth_packed_batch(rank: int, i: int, device):
    """[1, 16384] packed row: sequences of ~U{64..2048} tokens, 45 % of the ids from the unit range and 55 % from
    the text range, position_ids restarting at each sequence, labels -100 at starts."""
    g = torch.Generator().manual_seed(99 + rank + 1000 * i)
    total, ids, pos, lab = W4["tokens"], [], [], []
    while total > 0:
        n = min(total, int(torch.randint(64, W4["ctx"] + 1, (1,), generator=g)))
        unit = torch.rand(n, generator=g) < 0.45
        t = torch.where(unit, torch.randint(W4["unit_lo"], W4["vocab"], (n,), generator=g),
                        torch.randint(2, W4["unit_lo"], (n,), generator=g))
        t[0] = 1
        l = t.clone()
        l[0] = -100
        ids.append(t); pos.append(torch.arange(n)); lab.append(l)
        total -= n
    cat = lambda xs: torch.cat(xs)[None].to(device)  # noqa: E731
    lens = [len(x) for x in ids]
    return {"input_ids": cat(ids), "position_ids": cat(pos), "labels": cat(lab)}, lens


def w4_flops_per_batch(lens):
    """fwd+bwd algorithmic flops of one packed batch: 6 x matmul params per token + causal-exact attention."""
    L, H, nH, hd, I, Vp = 28, 1536, 12, 128, 8960, W4["vocab"]
    n_mm = L * (H * (nH + 4) * hd + nH * hd * H + 3 * H * I) + Vp * H
    attn = sum(L * 4 * nH * hd * (n * (n + 1) / 2) for n in lens)  # QK^T + PV, fwd
    return 6.0 * n_mm * sum(lens) + 3.0 * attn


def dominant_kernel_roofline(model, iters=50, warm=40):
    """gate|up projection forward GEMM of one layer at the bench shape, timed with HIP events on the
    stream it is launched on (torch's current stream)."""
    from slamkit_amd import engine as E
    lib = E.load_library()
    M, N, K = B * T, 2 * 4864, 896
    x = (torch.randn(M, K, device=model.device) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=model.device) * 0.02).to(torch.bfloat16)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=model.device)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=model.device)
    st = E.current_stream_ptr()
    for _ in range(warm):  # fresh 240 MB of outputs: the first few dozen launches run ~15 % slow (page / TLB warm-up)
        lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, N, K, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.slam_op_gemm_nt_swiglu(x.data_ptr(), w.data_ptr(), y.data_ptr(), act.data_ptr(), M, N, K, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * N * K
    ach = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "gemm_nt_256_kernel (NT, 256x256x64 tiles, 8 waves, 8-phase LDS-DMA schedule, persistent blocks) gate|up + fused SwiGLU, M8192 N9728 K896",
            "achieved": round(ach, 1),
            "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": round(ach * 1e12 / PEAK_BF16, 4),
            "ms_per_launch": round(ms, 4),
            # L2-miss-side bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes) of this
            # same persistent launch inside an optimizer step: a RECORDED measurement (profiles/r6_pmc_step.md, row
            # `gemm_nt_256_kernel<true, false, false> [256 blocks] fwd`: 231.1 MB read + 239.9 MB written; round 5: 233.3 + 239.6), NOT
            # collected by this run - counters need rocprofv3 around the process (tools/pmc_step.sh). The write side is exactly
            # algorithmic (159.4 MB gate|up + 79.7 MB act); the read side is 7.2x the 32 MB of operands: each of the 8 XCD-private
            # L2s streams the 17.4 MB weight (L2 misses, served by the 256 MB infinity cache after the first XCD: FETCH_SIZE is
            # not HBM reads).
            "traffic": 471.0e6, "traffic_source": "RECORDED, not measured by this run: profiles/r6_pmc_step.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over "
                                                  "tools/one_step.py, the same persistent launch, round-6 library)",
            "algorithmic_bytes": 271.2e6}


def kernel_rooflines(model, iters=30, warm=10):
    """The step's launch families by time (profiles/r4_step_breakdown.md), each timed live here as a STANDALONE launch
    at the bench shape with HIP events on its launch stream: algorithmic flops / mean launch time against the dense bf16
    MFMA peak. (Inside the step the weight-gradient GEMMs run as background launches beside the dgrad chain and every
    launch takes longer than alone; the in-step durations are `roofline.in_step`, measured live, and the rocprof summary.)"""
    from slamkit_amd import engine as E
    lib = E.load_library()
    st = E.current_stream_ptr()
    dev = model.device
    M, H, I, QKV = B * T, 896, 4864, 1152
    bf = lambda *s, sc=0.5: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)  # noqa: E731
    rows = []

    def add(name, flops, fn):
        us = _time_us(fn, iters=iters, warm=warm)
        rows.append({"kernel": name, "gflop": round(flops / 1e9, 1), "us": round(us, 1), "tflops": round(flops / us / 1e6, 1),
                     "frac": round(flops / (us * 1e-6) / PEAK_BF16, 4)})

    x, x2 = bf(M, H), bf(M, I)
    wgu, wd, wdt, wgut = bf(2 * I, H, sc=0.02), bf(H, I, sc=0.02), bf(I, H, sc=0.02), bf(H, 2 * I, sc=0.02)
    gu, act, y = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=dev), torch.empty(M, I, dtype=torch.bfloat16, device=dev), bf(M, H)
    dgu = bf(M, 2 * I, sc=0.1)
    dw_gu = torch.empty(2 * I, H, dtype=torch.float32, device=dev)
    dw_d = torch.empty(H, I, dtype=torch.float32, device=dev)
    ws = torch.empty(max(lib.slam_op_gemm_tn_workspace(M, 2 * I, H), lib.slam_op_gemm_tn_workspace(M, H, I)) // 4 + 16,
                     dtype=torch.float32, device=dev)
    p = lambda t: t.data_ptr()  # noqa: E731
    add("gate|up fwd + SwiGLU (NT 256x256 8-phase, persistent) M8192 N9728 K896", 2.0 * M * 2 * I * H,
        lambda: lib.slam_op_gemm_nt_swiglu(p(x), p(wgu), p(gu), p(act), M, 2 * I, H, st))
    add("gate|up wgrad, chip-filling plan (TN 128x128 balanced, contraction 8192 tokens) N9728 K896", 2.0 * M * 2 * I * H,
        lambda: lib.slam_op_gemm_tn(p(dgu), p(x), p(dw_gu), 0, M, 2 * I, H, p(ws), st))

    def as_in_step(fn):  # the plan the step uses on its weight-gradient stream: 256x224 tiles, one block per tile, no K-split
        def run():
            fn()
        return run
    for k, v in ((b"gemm_tn224", 2), (b"gemm_tn224_max_split", 1)):
        assert lib.slam_set_option(None, k, v) == 0
    try:
        add("gate|up wgrad AS LAUNCHED IN THE STEP (TN 256x224 8-phase, 152 one-per-CU blocks, no K-split) - largest launch family by time",
            2.0 * M * 2 * I * H, as_in_step(lambda: lib.slam_op_gemm_tn(p(dgu), p(x), p(dw_gu), 0, M, 2 * I, H, p(ws), st)))
        add("down wgrad AS LAUNCHED IN THE STEP (TN 256x224 8-phase, 76 blocks)", 2.0 * M * I * H,
            as_in_step(lambda: lib.slam_op_gemm_tn(p(y), p(x2), p(dw_d), 0, M, H, I, p(ws), st)))
    finally:
        lib.slam_set_option(None, b"gemm_tn224", 1)
        lib.slam_set_option(None, b"gemm_tn224_max_split", 16)
    add("gate|up dgrad (NT) M8192 N896 K9728", 2.0 * M * 2 * I * H,
        lambda: lib.slam_op_gemm_nt(p(dgu), p(wgut), p(y), None, None, M, H, 2 * I, 1, st))
    add("down fwd + residual (NT) M8192 N896 K4864", 2.0 * M * I * H,
        lambda: lib.slam_op_gemm_nt(p(x2), p(wd), p(y), None, p(x), M, H, I, 1, st))
    add("down dgrad + fused SwiGLU backward (NT 256x256) M8192 N4864 K896", 2.0 * M * I * H,
        lambda: lib.slam_op_gemm_nt_dswiglu(p(y), p(wdt), p(gu), M, I, H, st))
    add("down wgrad, chip-filling plan (TN 128x128 balanced) N896 K4864", 2.0 * M * I * H,
        lambda: lib.slam_op_gemm_tn(p(y), p(x2), p(dw_d), 0, M, H, I, p(ws), st))
    # attention at the bench shape: 8 sequences x 1024 tokens, 14 / 2 heads of 64; causal-exact flops
    nH, nKV, hd = 14, 2, 64
    qkv = bf(M, (nH + 2 * nKV) * hd, sc=1.0)
    o = torch.empty(M, nH * hd, dtype=torch.bfloat16, device=dev)
    do, dqkv = bf(M, nH * hd, sc=1.0), torch.empty(M, (nH + 2 * nKV) * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(nH * M, dtype=torch.float32, device=dev)
    ss = (torch.arange(M, device=dev, dtype=torch.int32) // T) * T
    se = ss + T
    aws = torch.empty(lib.slam_op_attn_bwd_workspace(M, nH, hd) // 4 + 16, dtype=torch.float32, device=dev)
    fl = 4.0 * hd * (T * (T + 1) / 2) * B * nH
    add("attention fwd (+ block-order plan) 8x1024, 14/2 heads of 64", fl,
        lambda: lib.slam_op_attn_fwd(p(qkv), p(o), p(lse), p(ss), M, nH, nKV, hd, st))
    add("attention bwd (plan + dQ + dK/dV + reduce; 7 matmuls, counted as 5)", 2.5 * fl,
        lambda: lib.slam_op_attn_bwd(p(qkv), p(o), p(do), p(lse), p(dqkv), p(aws), p(ss), p(se), M, nH, nKV, hd, st))
    return rows


def in_step_table(fam):
    """Launch families of the step as they run IN the step (HIP timing events around each launch on its own stream, three
    optimizer steps after the timed region): launches per step, mean us per launch, ms per step, and for the matmul families
    the algorithmic flops -> fraction of the dense bf16 peak at that in-step duration. Sorted by ms per step; the first row
    is the dominant family by time. Durations on the two backward streams overlap, so the column sums exceed the step."""
    M, H, I, QKV, HD, VP = B * T, 896, 4864, 1152, 896, 512
    attn = 4.0 * 64 * (T * (T + 1) / 2) * B * 14
    flops = {"qkv_fwd": 2.0 * M * QKV * H, "o_fwd": 2.0 * M * H * HD, "gateup_fwd": 2.0 * M * 2 * I * H, "down_fwd": 2.0 * M * H * I,
             "attn_fwd": attn, "attn_bwd": 2.5 * attn, "head_fwd": 2.0 * M * VP * H, "head_dgrad": 2.0 * M * VP * H, "head_wgrad": 2.0 * M * VP * H,
             "down_dgrad_dswiglu": 2.0 * M * H * I, "gateup_dgrad": 2.0 * M * 2 * I * H, "o_dgrad": 2.0 * M * H * HD, "qkv_dgrad": 2.0 * M * QKV * H,
             "wd_wgrad": 2.0 * M * H * I, "wgu_wgrad": 2.0 * M * 2 * I * H, "wo_wgrad": 2.0 * M * H * HD, "wqkv_wgrad": 2.0 * M * QKV * H,
             "embed_wgrad": 2.0 * M * VP * H}
    side = {"wd_wgrad", "wgu_wgrad", "wo_wgrad", "wqkv_wgrad", "head_wgrad", "embed_wgrad"}
    rows = []
    for name, v in fam.items():
        us = sum(v) / len(v) * 1e3
        r = {"family": name, "stream": "wgrad side stream" if name in side else "caller's stream", "launches_per_step": round(len(v) / 3, 1),
             "us_in_step": round(us, 1), "ms_per_step": round(sum(v) / 3, 3)}
        if name in flops:
            r["gflop"] = round(flops[name] / 1e9, 1)
            r["frac_of_peak_in_step"] = round(flops[name] / (us * 1e-6) / PEAK_BF16, 4)
        rows.append(r)
    rows.sort(key=lambda r: -r["ms_per_step"])
    # by KERNEL (one kernel serves several families): launch time per step and the flops those launches carry
    kernel_of = {"wd_wgrad": "gemm_tn_224_kernel (256x224 8-phase weight gradient)", "wgu_wgrad": "gemm_tn_224_kernel (256x224 8-phase weight gradient)",
                 "gateup_fwd": "gemm_nt_256_kernel (256x256 8-phase, persistent)", "down_dgrad_dswiglu": "gemm_nt_256_kernel (256x256 8-phase, persistent)",
                 "gateup_dgrad": "gemm_nt_224_kernel (256x224 8-phase)", "attn_bwd": "attn_bwd_dq / dkv / reduce kernels", "attn_fwd": "attn_fwd_kernel",
                 "wo_wgrad": "gemm_tn_bal_kernel + reduce (128x128 balanced weight gradient)", "wqkv_wgrad": "gemm_tn_bal_kernel + reduce (128x128 balanced weight gradient)",
                 "head_wgrad": "gemm_tn_bal_kernel + reduce (128x128 balanced weight gradient)", "embed_wgrad": "gemm_tn_bal_kernel + reduce (128x128 balanced weight gradient)"}
    by_kernel = {}
    for r in rows:
        k = kernel_of.get(r["family"], "gemm_kernel (128x128, fused epilogues)" if "gflop" in r else "memory-bound kernels (norms, loss)")
        e = by_kernel.setdefault(k, {"kernel": k, "ms_per_step": 0.0, "gflop_per_step": 0.0})
        e["ms_per_step"] += r["ms_per_step"]
        e["gflop_per_step"] += r.get("gflop", 0.0) * r["launches_per_step"]
    kernels = sorted(by_kernel.values(), key=lambda e: -e["ms_per_step"])
    for e in kernels:
        e["ms_per_step"] = round(e["ms_per_step"], 3)
        e["gflop_per_step"] = round(e["gflop_per_step"], 1)
        if e["gflop_per_step"]:
            e["frac_of_peak_in_step"] = round(e["gflop_per_step"] * 1e9 / (e["ms_per_step"] * 1e-3) / PEAK_BF16, 4)
    return {"dominant_by_time": rows[0]["family"] if rows else None,
            # the kernel with the most launch time per step and what it achieves THERE (queueing behind the other stream's
            # blocks included - on the side stream that is most of the difference to the stand-alone figure)
            "largest_kernel_by_time": kernels[0] if kernels else None, "kernels_by_time": kernels, "families": rows,
            "measured": "HIP timing-event pairs around every launch on its own stream (slam_family_ms), 3 optimizer steps after the timed region"}


_JSON_FD = None  # duplicate of the process's original stdout (main)


def emit(line: dict) -> None:
    """The ONE JSON line, on the original stdout."""
    data = (json.dumps(line) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


PEAK_HBM = 8.0e12  # B/s, MI355X_MICROARCH.md (about 6.3e12 reachable by a streaming copy)


def _time_us(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def hbm_kernel_rates(model, trainer):
    """The memory-bound kernels of the step at the bench shape: algorithmic bytes per launch
     / mean launch time measured here with HIP events -> GB/s against the 8 TB/s HBM peak. The matching
    counter-based figures (FETCH_SIZE / WRITE_SIZE per kernel) are in profiles/r2_pmc_step.md."""
    from slamkit_amd import engine as E
    lib = E.load_library()
    st = E.current_stream_ptr()
    dev = model.device
    M, H = B * T, 896
    x = (torch.randn(M, H, device=dev)).to(torch.bfloat16)
    w = torch.ones(H, dtype=torch.bfloat16, device=dev)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    rstd = torch.empty(M, dtype=torch.float32, device=dev)
    dw = torch.empty(H, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.slam_op_rmsnorm_bwd_workspace(M, H) // 4 + 16, dtype=torch.float32, device=dev)
    out = []

    def row(name, us, nbytes, what):
        out.append({"kernel": name, "us": round(us, 2), "algorithmic_bytes": int(nbytes), "GB_per_s": round(nbytes / us / 1e3, 1),
                    "frac_of_8TBps": round(nbytes / (us * 1e-6) / PEAK_HBM, 3), "bytes": what})

    us = _time_us(lambda: lib.slam_op_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), rstd.data_ptr(), M, H, 1e-6, st))
    row("rmsnorm_fwd_kernel [8192 x 896]", us, 4 * M * H, "4 B/elem: bf16 read + bf16 write")
    # (dw = NULL: per-block weight-gradient partials only, as the engine launches it - the column sums of a whole
    #  gradient bucket are finished in one batched launch)
    us = _time_us(lambda: lib.slam_op_rmsnorm_bwd(y.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), x.data_ptr(), dx.data_ptr(),
                                                   None, ws.data_ptr(), M, H, st))
    row("rmsnorm_bwd_kernel [8192 x 896] (+ fused residual-gradient add)", us, 8 * M * H, "8 B/elem: x, dy, dres read + dx written")
    n = model.engine.n_params
    eng = model.engine
    if trainer.state_dtype == torch.float32:
        us = _time_us(lambda: eng.adamw_step(model.flat_master, trainer.exp_avg, trainer.exp_avg_sq, trainer.norm_out, 0.0, 0.9, 0.999,
                                             1e-8, 0.0, 1000, zero_grad=False), iters=5, warm=2)
        # (the launch includes the transposed-weight-image refresh: + 4 B per matrix element)
        row("adamw_tile_kernel (fp32 master + moments; writes the transposed bf16 images itself)", us, 32 * n,
            "30 B/param AdamW + 2 B/param transposed image")
    else:
        us = _time_us(lambda: eng.adamw_step_bf16(trainer.exp_avg, trainer.exp_avg_sq, trainer.norm_out, 0.0, 0.9, 0.999, 1e-8, 0.0, 1000,
                                                  zero_grad=False), iters=5, warm=2)
        g16 = getattr(trainer, "_final_mode", 0) == 2
        row("adamw_tile_kernel (bf16 parameters + bf16 moments in place, the recipe's precision; writes the transposed bf16 images itself)",
            us, (16 if g16 else 18) * n, ("14 B/param AdamW (bf16 grad, p, m, v read; p, m, v written)" if g16 else
                                          "16 B/param AdamW (fp32 grad read, bf16 p/m/v read + written)") + " + 2 B/param transposed image")
    us = _time_us(lambda: eng.grad_norm(0.5, trainer.norm_out), iters=10, warm=2)
    if getattr(trainer, "_final_mode", 0) and getattr(trainer.args, "grad_norm_from_backward", True) and not trainer.reducer.force:
        # round 6: no pass over the gradients is left - backward's final-value stores emit ~60 k per-block sums of squares,
        # this launch adds them (one block, fp64): latency, not bandwidth
        out.append({"kernel": "norm_finish_kernel (global gradient norm from the partial sums backward's final-value stores emit)", "us": round(us, 2),
                    "algorithmic_bytes": None, "GB_per_s": None, "frac_of_8TBps": None,
                    "bytes": "~0.25 MB of partial sums; the 4 B/param pass of rounds 1-5 (0.245 ms) is gone"})
    else:
        row("sumsq_chunks_kernel + norm_finish_kernel (global gradient norm)", us, 4 * n, "4 B/param")
    return out


def extra_measurements(model, trainer, rank, dev, a):
    """After the timed region (never part of `value`): the same step at GA = 16 (the optimizer and the exposed gradient
    exchange amortised over 16 micro-batches) and with the recipe's own optimizer precision
    (bf16 parameters + bf16 Adam moments, /root/reference config/model/slam.yaml:9)."""
    from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments
    res = {}
    world = dist.get_world_size() if dist.is_initialized() else 1

    def run(tr, ga, steps, warm):
        bt = [[synth_batch(rank, 100 + j, dev) for j in range(ga)]]
        n_items = float(B * T * ga)
        for _ in range(warm):
            tr.optimizer_step(bt[0], 1e-3, counts=(n_items, n_items))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.optimizer_step(bt[0], 1e-3, counts=(n_items, n_items))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"tokens_per_s": round(world * B * T * ga * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps}

    res["grad_accum_16"] = run(trainer, 16, 3, 1)
    res.update(recipe_data_modes(model, trainer, rank, dev, world))
    # the boundary as the training loop crosses it: rows collated on the host (DataCollatorForLanguageModeling), CPU int64
    # batches handed to UnitLM.forward (pageable H2D inside the step), token counts taken from the CPU labels
    from slamkit_amd.data import DataCollatorForLanguageModeling
    coll = DataCollatorForLanguageModeling(pad_token_id=0)
    g = torch.Generator().manual_seed(77 + rank)
    rows = [[{"input_ids": [1] + torch.randint(2, V, (T - 1,), generator=g).tolist(), "attention_mask": [1] * T} for _ in range(B)]
            for _ in range(4)]
    steps_h = max(5, min(a.steps, 20))
    for i in range(3):
        trainer.optimizer_step([coll(rows[i % 4])], 1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps_h):
        trainer.optimizer_step([coll(rows[i % 4])], 1e-3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["host_boundary"] = {"tokens_per_s": round(world * B * T * steps_h / dt, 1), "ms_per_step": round(dt / steps_h * 1e3, 3), "steps": steps_h,
                            "what": "collate on the host + CPU int64 batches through UnitLM.forward (H2D inside the step)"}
    other = "bfloat16" if trainer.state_dtype == torch.float32 else "float32"
    args2 = SLAMTrainingArguments(per_device_train_batch_size=B, gradient_accumulation_steps=1, learning_rate=1e-3, max_grad_norm=0.5,
                                  logging_steps=0, optim_state_dtype=other,
                                  ddp_comm_dtype=os.environ.get("SLAM_DDP_COMM_DTYPE", "bfloat16"))
    del trainer.exp_avg, trainer.exp_avg_sq
    if other == "float32":  # the headline ran in the recipe's precision: give the model an fp32 master copy for this leg
        model.flat_master = model.flat_params.float()
        tr2 = SLAMTrainer(model=model, args=args2)
        res["fp32_master_optimizer"] = dict(run(tr2, 1, max(5, min(a.steps, 20)), 3),
                                            what="AdamW with fp32 master weights + fp32 moments (30 B/param) instead of the recipe's bf16 state")
    else:
        tr2 = SLAMTrainer(model=model, args=args2)  # drops the fp32 master: the bf16 parameters become the only copy
        res["recipe_optimizer_bf16_state"] = run(tr2, 1, max(5, min(a.steps, 20)), 3)
    return res


def recipe_data_modes(model, trainer, rank, dev, world):
    """The recipe's own data modes at Slam-358M, through SLAMTrainer.optimizer_step (VERDICT r5 missing #3): `packed_ga16` -
    data.packing=true with gradient_accumulation_steps=16 (/root/reference README.md:89), one flattened [1, 8192] row per
    micro-batch; `padded` - right-padded [8, 1024] rows (the non-packing collator). tokens_per_s counts non-ignored labels
    (SLAMTrainer.get_num_tokens, slam_trainer.py:59-65); positions_per_s counts every position the kernels process. The
    attention kernels see short segments / ragged rows here; their in-step durations come from one instrumented step."""
    res = {}

    def measure(name, micro_sets, steps, warm, what):
        counts = [float(sum(int((mb["labels"] != -100).sum()) for mb in ms)) for ms in micro_sets]
        positions = [sum(int(mb["input_ids"].numel()) for mb in ms) for ms in micro_sets]
        for i in range(warm):
            trainer.optimizer_step(micro_sets[i % len(micro_sets)], 1e-3, counts=(counts[i % len(counts)],) * 2)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tok = pos = 0
        for i in range(steps):
            k = i % len(micro_sets)
            trainer.optimizer_step(micro_sets[k], 1e-3, counts=(counts[k],) * 2)
            tok += counts[k]
            pos += positions[k]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # one more step with timing events around every launch: the attention kernels on THIS data
        model.engine.set_option("time_families", 1)
        trainer.optimizer_step(micro_sets[0], 1e-3, counts=(counts[0],) * 2)
        fam = {}
        for nm, ms_ in model.engine.family_ms():   # the marks of the step's LAST micro-batch
            fam.setdefault(nm, []).append(ms_)
        model.engine.set_option("time_families", 0)
        us = lambda k: round(sum(fam.get(k, [0.0])) / max(1, len(fam.get(k, [0.0]))) * 1e3, 1)  # noqa: E731
        res[name] = {"tokens_per_s": round(world * tok / dt, 1), "positions_per_s": round(world * pos / dt, 1),
                     "ms_per_step": round(dt / steps * 1e3, 3), "ms_per_micro_batch": round(dt / steps / len(micro_sets[0]) * 1e3, 3),
                     "steps": steps, "micro_batches_per_step": len(micro_sets[0]),
                     "in_step_us": {"attn_fwd": us("attn_fwd"), "attn_bwd": us("attn_bwd"), "qkv_fwd": us("qkv_fwd"), "gateup_fwd": us("gateup_fwd"),
                                    "loss": us("loss")},
                     "what": what}

    packed = [[synth_packed_358m(rank, 300 + 16 * s_ + j, dev) for j in range(16)] for s_ in range(2)]
    seg = [n for ms in packed for _, lens in ms for n in lens]
    measure("packed_ga16", [[mb for mb, _ in ms] for ms in packed], 3, 1,
            f"data.packing=true, GA 16 (README.md:89): flattened [1, 8192] rows, segment lengths U{{64..1024}} (mean {sum(seg) / len(seg):.0f}), "
            "labels -100 at segment starts, seed 1234 + rank")
    padded = [[synth_padded_358m(rank, 400 + s_, dev)] for s_ in range(4)]
    fill = sum(n for ms in padded for _, lens in ms for n in lens) / (len(padded) * B * T)
    measure("padded", [[mb for mb, _ in ms] for ms in padded], 10, 3,
            f"right-padded [8, 1024] rows, lengths U{{256..1024}} ({100 * fill:.0f} % of the positions are tokens), labels -100 on the padding, seed 4321")
    return res


def dp_variant_table(model, trainer, args, rank, world, dev, rows):
    """N > 1 only, after the timed region (which stays ddp_algo = rs_ag, bf16 exchange, 4 layers per bucket): the one driver
    run at N GPUs also decides the open data-parallel questions - 5 optimizer steps (after 3 warm-up steps)
    of every exchange variant on the same model: the median step time (max over ranks) and the exposed communication time of
    the last step. Every variant is a fresh SLAMTrainer (its own reducer, bucket plan and optimizer state) on the same engine."""
    import dataclasses
    from slamkit_amd.trainer import SLAMTrainer
    variants = [("rs_ag", "bfloat16", 4), ("all_reduce", "bfloat16", 4), ("rs_ag", "float32", 4), ("rs_ag", "bfloat16", 2), ("rs_ag", "bfloat16", 8),
                ("all_reduce", "bfloat16", 8),
                # the same exchange with the persistent 256 x 256 grids on 240 / 224 of the 256 CUs: do RCCL's kernels need CUs of their own?
                ("rs_ag", "bfloat16", 4, 240), ("rs_ag", "bfloat16", 4, 224)]
    n_items = float(B * T)
    batch = [synth_batch(rank, 200, dev)]
    # ONE communication stream for every variant: each reducer would otherwise create its own, and streams beyond the HIP
    # hardware queues (GPU_MAX_HW_QUEUES = 8) share a queue with the compute or the weight-gradient stream - measured on a
    # 1-rank group: two of six variants 12 ms per step slower for no other reason
    shared_side = trainer.reducer.side
    for algo, cd, bl, *rest in variants:
        pcus = rest[0] if rest else 0
        model.engine.set_option("gemm_256_persist_cus", pcus)
        model.engine.join()
        torch.cuda.synchronize()
        trainer.exp_avg = trainer.exp_avg_sq = None  # one optimizer state at a time
        a2 = dataclasses.replace(args, ddp_algo=algo, ddp_comm_dtype=cd, ddp_bucket_layers=bl, gradient_accumulation_steps=1)
        tr = SLAMTrainer(model=model, args=a2)
        tr.reducer.side = shared_side
        nv = int(os.environ.get("SLAM_BENCH_DP_VARIANT_STEPS", "5"))
        for _ in range(min(3, nv)):  # the first steps of a fresh reducer allocate its staging buffers and set RCCL up for its message sizes
            tr.optimizer_step(batch, 1e-3, counts=(n_items, n_items))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(nv + 1)]
        ev[0].record()
        for i in range(nv):
            tr.optimizer_step(batch, 1e-3, counts=(n_items, n_items))
            ev[i + 1].record()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(nv))[nv // 2]  # median step (this rank's stream), MAX over ranks below
        ex = tr.reducer.exposed_ms()
        if world > 1:
            t = torch.tensor([dt, ex], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, ex = float(t[0]), float(t[1])
        tr._gather_optimizer_state()
        rows.append({"algo": algo, "comm_dtype": cd, "bucket_layers": bl, "persistent_grid_cus": pcus or "all", "ms_per_step": round(dt, 3),
                     "exposed_comm_ms": round(ex, 3), "tokens_per_s": round(world * B * T / (dt * 1e-3), 1)})
        trainer = tr
    model.engine.set_option("gemm_256_persist_cus", 0)


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (a container
    that reports 256 CPUs but is quota-limited to 8 must not spin 256 OpenMP threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_worker(steps: int, seq: int):
    """fp32 oracle step (fwd + shifted CE + bwd + clip + AdamW), Slam-358M, on the host cores."""
    from oracle import slam_oracle as O
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    cfg = O.SLAM_358M
    g = torch.Generator().manual_seed(0)
    sd = {k: (torch.ones(s) if k.endswith("norm.weight") else torch.zeros(s) if k.endswith(".bias")
              else torch.randn(*s, generator=g) * 0.02) for k, s in O.hf_keys(cfg)}
    ids = torch.randint(2, V, (1, seq), generator=torch.Generator().manual_seed(1234))
    ids[:, 0] = 1
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(v) for k, v in sd.items()}
    times = []
    for s in range(steps + 1):
        t0 = time.time()
        _, _, grads = O.forward_loss_grads(cfg, sd, ids, ids)
        _, coef = O.clip_coef(grads, 0.5)
        for k in sd:
            O.adamw_update(sd[k], grads[k] * coef, m[k], v[k], s + 1, 1e-3)
        times.append(time.time() - t0)
    dt = sorted(times[1:])[len(times[1:]) // 2]
    print(json.dumps({"value": round(seq / dt, 1), "unit": "tokens/s", "cores": cores, "kind": "port",
                      "sample": f"oracle fp32 fwd+CE+bwd+clip+AdamW, Slam-358M, B=1 T={seq}, median of {steps} "
                                f"step(s) after 1 warm-up", "sec_per_step": round(dt, 2)}), flush=True)


def cpu_baseline(timeout_s=240):
    """Runs the worker in a subprocess with a hard wall-clock bound so the default bench stays short."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True,
                           text=True, timeout=timeout_s, env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "tokens/s", "cores": usable_cores(), "kind": "port",
                "sample": "worker produced no result: " + (r.stderr or "")[-200:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "tokens/s", "cores": usable_cores(), "kind": "port",
                "sample": f"oracle step did not finish within {timeout_s}s on this host"}


def spawn_ranks(n: int) -> None:
    """Re-run this script under torch.distributed.run with n ranks on this node (127.0.0.1 rendezvous, free port)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node")
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    if _JSON_FD is not None:  # the launched ranks inherit the real stdout (rank 0 prints the line there)
        os.dup2(_JSON_FD, 1)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def bench_qwen1p5b(a, world, rank, dev):
    """Extra workload (not the BASELINE metric): same step loop on the configs[3]-shaped model."""
    from slamkit_amd.model import UnitLM, UnitLMConfig
    from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments
    cfg = UnitLMConfig(base_model_name=W4["name"], vocab_size=W4["vocab"], max_tokens=W4["tokens"])
    model = UnitLM(cfg, seed=0)
    args = SLAMTrainingArguments(per_device_train_batch_size=1, gradient_accumulation_steps=1, learning_rate=5e-4,
                                 max_grad_norm=0.5, logging_steps=0,
                                 optim_state_dtype=os.environ.get("SLAM_OPTIM_STATE_DTYPE", "bfloat16"))
    trainer = SLAMTrainer(model=model, args=args)
    nb = 4
    made = [synth_packed_batch(rank, i, dev) for i in range(nb)]
    batches = [[m[0]] for m in made]
    counts = [float((m[0]["labels"] != -100).sum()) for m in made]

    def step(i):
        trainer.optimizer_step(batches[i % nb], 5e-4, counts=(counts[i % nb], counts[i % nb]))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        toks = sum(counts[(a.warmup + i) % nb] for i in range(a.steps))
        flops = sum(w4_flops_per_batch(made[(a.warmup + i) % nb][1]) for i in range(a.steps))
        emit(({
            "metric": "train tokens/sec (whole node), Qwen2.5-1.5B-shaped interleaved model ctx=2048 packed",
            "value": round(world * toks / dt, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[3]-shaped: 28 L, H 1536, 12/2 heads of 128, I 8960, vocab 152167, 16384 packed "
                                   "tokens per micro-batch, random-init weights; full optimizer step (AdamW " + args.optim_state_dtype + " state)",
                       "parallelism": f"dp{world}", "final_loss": round(float(trainer._loss_acc) / max(1, trainer._loss_n), 4)},
            "roofline": {"bound": "mfma", "step_tflops_per_gpu": round(flops / dt / 1e12, 1),
                         "step_frac": round(flops / dt / PEAK_BF16, 4), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s"},
        }))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def bench_dpo(a, world, rank, dev):
    """Extra workload (not the BASELINE metric): BASELINE.json configs[4] / SURVEY.md §8d config 5 - the DPO step of
    cli/preference_alignment_train.py on Slam-358M: 8 preference pairs per GPU (prompt ~U{25..75} units, chosen / rejected
    ~U{50..150}), policy forward + backward over the 16 sequences, no-gradient reference forward, beta 0.1, clip + AdamW."""
    from slamkit_amd.model import UnitLM, UnitLMConfig
    from slamkit_amd.trainer import DPOConfig, SLAMDPOTrainer
    cfg = dict(base_model_name="Qwen/Qwen2.5-0.5B", rope_theta=10000.0, vocab_size=V, max_tokens=16 * 256)
    model = UnitLM(UnitLMConfig(**cfg), seed=0)
    ref = UnitLM(UnitLMConfig(**cfg), seed=0, allocate_grads=False)

    class _Tok:  # rows below are already token ids
        bos_token_id = eos_token_id = 1
        def __call__(self, s, add_special_tokens=False):
            return {"input_ids": list(s)}
    g = torch.Generator().manual_seed(4321 + rank)
    def ids(lo, hi):
        return torch.randint(2, V, (int(torch.randint(lo, hi + 1, (1,), generator=g)),), generator=g).tolist()
    pairs = [[{"prompt": ids(25, 75), "chosen": ids(50, 150), "rejected": ids(50, 150)} for _ in range(8)] for _ in range(4)]
    args = DPOConfig(per_device_train_batch_size=8, learning_rate=5e-5, max_grad_norm=0.5, logging_steps=0, beta=0.1,
                     optim_state_dtype=os.environ.get("SLAM_OPTIM_STATE_DTYPE", "bfloat16"))
    tr = SLAMDPOTrainer(model=model, ref_model=ref, args=args, train_dataset=[r for b in pairs for r in b], processing_class=_Tok())
    batches = [tr._collate_pairs(tr.train_dataset[8 * i: 8 * i + 8]) for i in range(4)]
    toks = [int((b["labels"] != -100).sum()) for b in batches]
    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for i in range(a.warmup):
        tr.optimizer_step([batches[i % 4]], 5e-5)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        tr.optimizer_step([batches[(a.warmup + i) % 4]], 5e-5)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        done = sum(toks[(a.warmup + i) % 4] for i in range(a.steps))
        emit(({
            "metric": "DPO preference pairs/sec (whole node), Slam-358M", "value": round(world * 8 * a.steps / dt, 1), "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[4]: DPO on Slam-358M, 8 pairs / GPU / step (16 sequences padded to a multiple of 64 tokens), policy "
                                   "fwd+bwd + reference fwd, beta 0.1; full optimizer step (AdamW " + args.optim_state_dtype + " state)", "parallelism": f"dp{world}",
                       "completion_tokens_per_s": round(world * done / dt, 1), "tokens_per_batch": [int(b["input_ids"].numel()) for b in batches],
                       "final_loss": round(float(tr._loss_acc) / max(1, tr._loss_n), 4)}}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # SURVEY.md §8d protocol: discard >= 10 steps, median of >= 50
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the GA=16 and recipe-optimizer measurements after the timed region")
    ap.add_argument("--grad-accum", type=int, default=1)
    ap.add_argument("--workload", default="slam358m", choices=["slam358m", "qwen1p5b", "dpo"],
                    help="slam358m = BASELINE.json configs[1] (the headline metric); qwen1p5b = configs[3]-shaped extra; dpo = configs[4] extra")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        cpu_baseline_worker(steps=3, seq=1024)
        return

    # stdout carries exactly ONE line - the JSON: everything else this process (or a library under it: gloo announces its
    # connections on stdout) prints goes to stderr; the line itself is written to the saved descriptor at the end
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one process per GPU, the reference's own launch model:
        # /root/reference cli/train.py:51,61 reads WORLD_SIZE / RANK set by torchrun, README.md:89)
        return spawn_ranks(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"[bench] --gpus {a.gpus} but the launcher started WORLD_SIZE={world}: measuring {world} rank(s)", file=sys.stderr)
    # SLAM_BENCH_DEVICE / SLAM_BENCH_BACKEND: plumbing check of the N > 1 path on a one-GPU box (all ranks on one device,
    # exchange over gloo) - tools/gpu_run.sh; a measurement run never sets them
    local = int(os.environ.get("SLAM_BENCH_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if os.environ.get("SLAM_BENCH_STREAM", "0") == "1" or int(os.environ.get("SLAM_BWD_WGRAD_CUS", "0")) > 0:
        # run the step on a non-default (non-blocking) stream: a CU-masked wgrad stream is a BLOCKING stream that would
        # otherwise synchronise with every launch on the NULL stream (include/slam_engine.h, "bwd_wgrad_cus")
        torch.cuda.set_stream(torch.cuda.Stream())
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a hung collective must fail the run in minutes, not after RCCL's default 10-minute watchdog per collective
        backend = os.environ.get("SLAM_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get("SLAM_NCCL_TIMEOUT_S", "180"))),
                                **({"device_id": dev} if backend == "nccl" else {}))
        # per-rank RCCL sanity line (stderr: stdout carries the ONE JSON line): a one-element all-reduce over the group
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"[bench] rank {dist.get_rank()}/{dist.get_world_size()} on cuda:{local} "
              f"({torch.cuda.get_device_name(local)}): {backend} all-reduce of ones = {float(t):.0f}", file=sys.stderr, flush=True)
        assert float(t) == float(dist.get_world_size())

    from slamkit_amd.model import UnitLM, UnitLMConfig
    from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments

    if a.workload == "qwen1p5b":
        return bench_qwen1p5b(a, world, rank, dev)
    if a.workload == "dpo":
        return bench_dpo(a, world, rank, dev)

    cfg = UnitLMConfig(base_model_name="Qwen/Qwen2.5-0.5B", rope_theta=10000.0, vocab_size=V, max_tokens=B * T)
    model = UnitLM(cfg, seed=0)
    args = SLAMTrainingArguments(per_device_train_batch_size=B, gradient_accumulation_steps=a.grad_accum,
                                 learning_rate=1e-3, max_grad_norm=0.5, logging_steps=0,
                                 overlap_optimizer=os.environ.get("SLAM_OVERLAP_OPTIMIZER", "0") == "1",
                                 overwrite_first_grad=os.environ.get("SLAM_OVERWRITE_FIRST_GRAD", "1") == "1",
                                 ddp_comm_dtype=os.environ.get("SLAM_DDP_COMM_DTYPE", "bfloat16"),
                                 # N > 1: reduce-scatter + sharded AdamW + parameter all-gather (optimizer time / N, one staging
                                 # pass); SLAM_DDP_ALGO=all_reduce selects the replicated update
                                 ddp_algo=os.environ.get("SLAM_DDP_ALGO", "rs_ag" if world > 1 else "all_reduce"),
                                 # the recipe's own optimizer precision (/root/reference config/model/slam.yaml:9 torch_dtype bfloat16:
                                 # the HF Trainer builds AdamW on bf16 parameters, so the moments are bf16 too); the fp32-master
                                 # variant is measured after the timed region (`extras.fp32_master_optimizer`)
                                 optim_state_dtype=os.environ.get("SLAM_OPTIM_STATE_DTYPE", "bfloat16"),
                                 # precision of the step's FINAL gradients: the recipe's own by default (bf16 parameters have
                                 # bf16 .grad); SLAM_GRAD_DTYPE=float32 for the A/B
                                 grad_dtype=os.environ.get("SLAM_GRAD_DTYPE") or None)
    trainer = SLAMTrainer(model=model, args=args)
    if os.environ.get("SLAM_FINAL_MODE") is not None:   # A/B only: 0 = the round-5 step (fp32 gradients + chunked norm pass)
        trainer._final_mode = int(os.environ["SLAM_FINAL_MODE"])
    nb = 4
    batches = [[synth_batch(rank, i * a.grad_accum + j, dev) for j in range(a.grad_accum)] for i in range(nb)]
    n_items = float(B * T * a.grad_accum)   # HF num_items_in_batch counts unshifted labels != -100
    trained_tokens = B * T * a.grad_accum  # SLAMTrainer.get_num_tokens definition: labels != -100 (slam_trainer.py:59-65)

    # the token-count all-reduce of step k + 1 is posted (host-side gloo, asynchronous) before step k is enqueued, as
    # SLAMTrainer.train() does with its collate-ahead batches: every step still has its own collective
    ahead = {"h": trainer.post_counts(n_items, n_items)}

    def step(i):
        h, ahead["h"] = ahead["h"], trainer.post_counts(n_items, n_items)
        trainer.optimizer_step(batches[i % nb], 1e-3, counts=(n_items, n_items), counts_handle=h)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]  # per-step device times (no host sync)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        step(a.warmup + i)
        marks[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    ms_median = per_step[len(per_step) // 2]
    loss = float(trainer._loss_acc) / max(1, trainer._loss_n)
    exposed = trainer.reducer.exposed_ms()  # last step: compute-stream stall behind the gradient all-reduce
    if world > 1:
        t = torch.tensor([exposed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed = float(t)
        # every rank accumulated local_sum / GLOBAL token count: the global token-mean loss is the sum over ranks
        t = torch.tensor([loss], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        loss = float(t)

    # SURVEY.md §8d protocol (discard >= 10 steps, median of >= 50): whatever K the launcher asked for, 50 more optimizer steps
    # follow the timed region, each between two HIP events; their median is `config.ms_per_step_median_50` (this rank)
    n50 = int(os.environ.get("SLAM_BENCH_MEDIAN_STEPS", "50"))  # (the one-GPU gloo plumbing check sets a few: its steps take seconds)
    marks50 = [torch.cuda.Event(enable_timing=True) for _ in range(n50 + 1)]
    marks50[0].record()
    for i in range(n50):
        step(a.warmup + a.steps + i)
        marks50[i + 1].record()
    fence()
    per50 = sorted(marks50[i].elapsed_time(marks50[i + 1]) for i in range(n50))
    ms_median_50 = per50[n50 // 2]
    base_i = a.warmup + a.steps + n50
    # after the timed region, on EVERY rank: the kernel probes touch the optimizer state (identically on all ranks), the
    # extra steps contain the data-parallel collectives
    # The dominant kernel where it runs: three more optimizer steps with timing events around every gate|up projection
    # launch (slam_gateup_launch_ms) - 72 launches between their real neighbours, the launches rocprofv3 reports for the step.
    # (A stand-alone loop of the same launch reads 137-162 us from run to run: 50 back-to-back launches of the
    # hottest kernel move with the power state of the part, profiles/r3_experiments/README.md, profiles/r4_vendor_gemm.md.)
    model.engine.set_option("time_gateup", 1)
    model.engine.set_option("time_param_waits", 1)  # rs_ag: the stall of the forward behind the parameter all-gather
    dp_on = world > 1 or trainer.reducer.force  # SLAM_DP_FORCE=1: the collective path on a 1-rank group (tools/dp1_bench.sh)
    trainer.reducer.time_buckets = dp_on
    in_step_ms = []
    for i in range(3):
        step(base_i + i)
        in_step_ms += model.engine.gateup_launch_ms(24)
    model.engine.set_option("time_gateup", 0)
    # ... and three more with a timing-event pair around EVERY launch family, on the stream each launch goes to
    # (slam_family_ms): the in-step duration of the dgrad chain and of the weight-gradient GEMMs on the side stream
    model.engine.set_option("time_families", 1)
    fam = {}
    for i in range(3):
        step(base_i + 3 + i)
        for name, ms_ in model.engine.family_ms():
            fam.setdefault(name, []).append(ms_)
    model.engine.set_option("time_families", 0)
    param_gather_ms = model.engine.param_wait_ms()
    model.engine.set_option("time_param_waits", 0)
    bucket_ms = trainer.reducer.bucket_ms() if dp_on else []  # this rank's collectives of the last of those steps
    trainer.reducer.time_buckets = False
    hbm = hbm_kernel_rates(model, trainer)
    # the extras are single-GPU context for the headline (GA 16, host boundary, bf16 state); a multi-rank run measures `value` only
    extras = None if (a.no_extras or world > 1) else extra_measurements(model, trainer, rank, dev, a)
    out = None
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * trained_tokens * a.steps / dt
        out = {
            "metric": "train tokens/sec (whole node), Slam-358M ctx=1024", "value": round(value, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: Slam-358M (Qwen2.5-0.5B body, vocab 502, rope_theta 1e4), ctx=1024, "
                                   "synthetic unit-token stream, random-init weights; full optimizer step ("
                                   + ("AdamW on bf16 parameters with bf16 moments" if args.optim_state_dtype == "bfloat16" else "AdamW, state " + args.optim_state_dtype) + ")",
                       "model": "Slam-358M", "global_batch": world * B * a.grad_accum, "micro_batch": B, "seq_len": T,
                       "grad_accum": a.grad_accum, "parallelism": f"dp{world}",
                       "optimizer": ("AdamW fp32 master+moments" if args.optim_state_dtype == "float32" else
                                     "AdamW on bf16 parameters with bf16 moments - the recipe's own precision (reference config/model/slam.yaml:9); "
                                     "fp32-master variant in extras") + ", clip 0.5",
                       "final_loss": round(loss, 4),
                       "optimizer_state_dtype": args.optim_state_dtype,
                       "ms_per_step_median": round(ms_median, 3), "ms_per_step_min": round(per_step[0], 3),
                       "ms_per_step_median_50": round(ms_median_50, 3), "median_50_steps": n50,  # 50 further steps after the timed region (median of >= 50)
                       "tokens_per_s_median_50": round(trained_tokens / (ms_median_50 * 1e-3), 1),  # this rank
                       "tokens_per_s_median_step": round(trained_tokens / (ms_median * 1e-3), 1),  # this rank
                       "ddp_algo": args.ddp_algo if dp_on else None,
                       "exposed_comm_ms_last_step": round(exposed, 3),
                       # [what, offset, elements, ms on the communication stream] per bucket, rank 0, one step after the timed region
                       "bucket_comm_ms": bucket_ms,
                       # rs_ag: stall of the forwards behind the parameter all-gather, summed over the 3 instrumented steps
                       "exposed_param_gather_ms_3_steps": round(param_gather_ms, 3)},
        }
        roof = dominant_kernel_roofline(model)
        ms_in = sum(in_step_ms) / len(in_step_ms)
        flops_gu = 2.0 * (B * T) * (2 * 4864) * 896
        roof["standalone"] = {"ms_per_launch": roof["ms_per_launch"], "achieved": roof["achieved"], "frac": roof["frac"],
                              "what": "50 back-to-back launches on probe buffers after the run"}
        roof["ms_per_launch"] = round(ms_in, 4)
        roof["achieved"] = round(flops_gu / (ms_in * 1e-3) / 1e12, 1)
        roof["frac"] = round(flops_gu / (ms_in * 1e-3) / PEAK_BF16, 4)
        roof["measured"] = (f"mean of {len(in_step_ms)} launches inside 3 optimizer steps after the timed region, HIP timing events "
                            "around each launch on its stream (slam_gateup_launch_ms); min "
                            f"{min(in_step_ms) * 1e3:.1f} us, max {max(in_step_ms) * 1e3:.1f} us")
        roof["step_frac"] = round(value / world * FLOP_PER_TOKEN / PEAK_BF16, 4)
        roof["step_tflops_per_gpu"] = round(value / world * FLOP_PER_TOKEN / 1e12, 1)
        roof["kernels"] = kernel_rooflines(model)
        roof["in_step"] = in_step_table(fam)
        # the headline `frac` above is the gate|up projection (the kernel with the most flops and the best fraction); the kernel
        # with the most LAUNCH TIME per step sits next to it, not three levels down
        big = roof["in_step"]["largest_kernel_by_time"] or {}
        roof["dominant_by_time"] = {"kernel": big.get("kernel"), "ms_per_step": big.get("ms_per_step"), "frac": big.get("frac_of_peak_in_step"),
                                    "what": "sum of in-step launch durations of this kernel (side-stream launches overlap the caller's stream); frac = its flops / that time / peak"}
        out["roofline"] = roof
        out["hbm_kernels"] = hbm
        if extras is not None:
            out["extras"] = extras
            # the attention (and the other shape-dependent) kernels on the recipe's own data modes, next to the dense in-step table
            roof["in_step"]["recipe_data_modes_us"] = {k: extras[k]["in_step_us"] for k in ("packed_ga16", "padded") if k in extras}
            if "fp32_master_optimizer" in extras:  # like-for-like with rounds 1-3 (fp32 master + fp32 moments), first-class
                out["value_fp32_master_optimizer"] = extras["fp32_master_optimizer"]["tokens_per_s"]
    # N > 1 (or SLAM_DP_FORCE=1 on one rank): the exchange variants, LAST - the line above is complete without them. A variant
    # that fails or hangs (none of these collectives has met a second GPU from the authoring side) must not cost the run its
    # headline: errors are recorded per variant, and a watchdog emits the line as it stands after SLAM_BENCH_DP_VARIANT_BUDGET_S
    if dp_on and os.environ.get("SLAM_BENCH_DP_VARIANTS", "1") == "1":
        import threading
        rows = []

        def bail():
            if rank == 0:
                out.setdefault("extras", {})["dp_variants"] = rows + [{"error": "watchdog: the variant table did not finish in time"}]
                emit(out)
            os._exit(0)
        dog = threading.Timer(float(os.environ.get("SLAM_BENCH_DP_VARIANT_BUDGET_S", "120")), bail)
        dog.daemon = True
        dog.start()
        try:
            dp_variant_table(model, trainer, args, rank, world, dev, rows)
        except Exception as e:  # noqa: BLE001 - recorded on the line
            rows.append({"error": f"{type(e).__name__}: {e}"[:300]})
        dog.cancel()
        if rank == 0:
            out.setdefault("extras", {})["dp_variants"] = rows
            out["extras"]["nccl_env"] = {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO", "GPU_MAX_HW_QUEUES")}
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            del trainer, model
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline()
        emit(out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
