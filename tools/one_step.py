"""Three optimizer steps of the bench workload and nothing else (no timing events, no probes): the process that
tools/pmc_step.sh wraps in rocprofv3 --pmc. Usage: python tools/one_step.py [slam358m|slam358m_packed|slam358m_padded|qwen1p5b]
(slam358m_packed: the recipe's data mode - one flattened [1, 8192] row per micro-batch, GA 4 here; slam358m_padded: right-padded rows)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from slamkit_amd.model import UnitLM, UnitLMConfig  # noqa: E402
from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "slam358m"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if wl == "slam358m":
    model = UnitLM(UnitLMConfig(base_model_name="Qwen/Qwen2.5-0.5B", rope_theta=10000.0, vocab_size=bench.V, max_tokens=bench.B * bench.T), seed=0)
    tr = SLAMTrainer(model=model, args=SLAMTrainingArguments(per_device_train_batch_size=bench.B, learning_rate=1e-3, max_grad_norm=0.5, logging_steps=0,
                                                          optim_state_dtype=os.environ.get("SLAM_OPTIM_STATE_DTYPE", "bfloat16")))
    n = float(bench.B * bench.T)
    for i in range(3):
        tr.optimizer_step([bench.synth_batch(0, i, dev)], 1e-3, counts=(n, n))
elif wl in ("slam358m_packed", "slam358m_padded"):
    ga = 4 if wl.endswith("packed") else 1
    model = UnitLM(UnitLMConfig(base_model_name="Qwen/Qwen2.5-0.5B", rope_theta=10000.0, vocab_size=bench.V, max_tokens=bench.B * bench.T), seed=0)
    tr = SLAMTrainer(model=model, args=SLAMTrainingArguments(per_device_train_batch_size=bench.B, gradient_accumulation_steps=ga, learning_rate=1e-3,
                                                          max_grad_norm=0.5, logging_steps=0, optim_state_dtype="bfloat16"))
    mk = bench.synth_packed_358m if wl.endswith("packed") else bench.synth_padded_358m
    for i in range(3):
        micro = [mk(0, 10 * i + j, dev)[0] for j in range(ga)]
        c = float(sum(int((mb["labels"] != -100).sum()) for mb in micro))
        tr.optimizer_step(micro, 1e-3, counts=(c, c))
else:
    model = UnitLM(UnitLMConfig(base_model_name=bench.W4["name"], vocab_size=bench.W4["vocab"], max_tokens=bench.W4["tokens"]), seed=0)
    tr = SLAMTrainer(model=model, args=SLAMTrainingArguments(per_device_train_batch_size=1, learning_rate=5e-4, max_grad_norm=0.5, logging_steps=0))
    for i in range(3):
        mb, _ = bench.synth_packed_batch(0, i, dev)
        c = float((mb["labels"] != -100).sum())
        tr.optimizer_step([mb], 5e-4, counts=(c, c))
torch.cuda.synchronize()
print("done", float(tr._loss_acc) / max(1, tr._loss_n))
