"""Socket power and shader clock of the forward alone, of forward + backward, and of the whole step (rocm-smi polled beside a
loop of each, ~12 s per phase): which phases of the Slam-358M step sit at the board's power limit?
  python tools/probes/phase_power.py"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402
from slamkit_amd.model import UnitLM, UnitLMConfig  # noqa: E402
from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = UnitLM(UnitLMConfig(base_model_name="Qwen/Qwen2.5-0.5B", rope_theta=10000.0, vocab_size=bench.V, max_tokens=bench.B * bench.T), seed=0)
tr = SLAMTrainer(model=model, args=SLAMTrainingArguments(per_device_train_batch_size=bench.B, learning_rate=1e-3, max_grad_norm=0.5,
                                                          logging_steps=0, optim_state_dtype="bfloat16"))
mb = bench.synth_batch(0, 0, dev)
n = float(bench.B * bench.T)


def poll(out, stop):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        p = re.search(r"Power \(W\): ([0-9.]+)", r)
        c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", r)
        if p and c:
            out.append((float(p.group(1)), int(c.group(1))))
        time.sleep(0.3)


def phase(name, fn, secs=12.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    out, stop = [], threading.Event()
    t0 = time.perf_counter()
    it = 0
    th = None
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        it += 20
        if th is None and time.perf_counter() - t0 > 3.0:   # warm: start polling
            th = threading.Thread(target=poll, args=(out, stop)); th.start()
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    pw = sorted(x[0] for x in out); ck = sorted(x[1] for x in out)
    print(f"{name:34s} {dt / it * 1e3:7.2f} ms per iteration | power W min {pw[0]:.0f} median {pw[len(pw) // 2]:.0f} max {pw[-1]:.0f} | "
          f"sclk MHz median {ck[len(ck) // 2]} ({len(out)} samples)", flush=True)


def fwd():
    model.forward(input_ids=mb["input_ids"], labels=mb["labels"], num_items_in_batch=n, return_logits=False)


def fwd_bwd():
    model.engine.set_option("grad_overwrite_next", 1)
    fwd()
    model.backward(1.0)


phase("forward only", fwd)
phase("forward + backward", fwd_bwd)
phase("whole optimizer step", lambda: tr.optimizer_step([mb], 1e-3, counts=(n, n)))
model.engine.set_option("bwd_wgrad_stream", 0)
phase("forward + backward, ONE stream", fwd_bwd)
