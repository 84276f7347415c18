"""Training-trajectory anchor (round 4): 200 optimizer steps of the REAL reference model on the HF/torch step.

The reference's `SLAMTrainer` cannot be constructed on this container's transformers 5.x (slam_trainer.py:50 passes
`tokenizer=`), so the Trainer-level step used to rest on the oracle's restatement alone. This script runs the pieces the
HF Trainer composes - unchanged, imported, not restated:
    slamkit.model.UnitLM (/root/reference slamkit/model/unit_lm.py:82-182, `compute_loss` :13-29 in its
        `num_items_in_batch` form) over a locally built Qwen2Config,
    torch.optim.AdamW (betas 0.9 / 0.999, eps 1e-8, wd 0: training_args.py defaults; fused on the bf16 leg),
    transformers.get_scheduler("cosine_with_min_lr", scheduler_specific_kwargs={"min_lr": 5e-5})
        (/root/reference config/training_args/default.yaml),
    torch.nn.utils.clip_grad_norm_(0.5) (default.yaml max_grad_norm),
in the loop order of Trainer._inner_training_loop (forward, backward, clip, optimizer.step, scheduler.step, zero_grad),
on the learnable token stream of tests/test_gpu_train.py::test_loss_curve_200_steps_vs_oracle (same rows, same seeded batch
order, same initial weights), in two precisions:
    fp32 : fp32 parameters, no autocast;
    bf16 : the recipe's own precision (/root/reference config/model/slam.yaml:9 torch_dtype bfloat16 + training_args bf16):
           bf16 parameters, bf16 autocast, AdamW state in bf16.
Run in the authoring container:
    HF_HUB_OFFLINE=1 python tests/golden/make_golden_traj.py
Writes tests/golden/traj.npz: per-step loss, learning rate and pre-clip gradient norm of both legs, and the L2 norm of
every final parameter tensor. tests/test_oracle_golden.py holds the oracle's step restatement to the fp32 leg and its
bf16 emulation to the bf16 leg; tests/test_gpu_train.py holds the engine to the same curves."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
stub = tempfile.mkdtemp()
os.makedirs(os.path.join(stub, "omegaconf"))
with open(os.path.join(stub, "omegaconf", "__init__.py"), "w") as f:
    f.write("class DictConfig(dict): pass\nclass ListConfig(list): pass\nclass OmegaConf: pass\n")
sys.path[:0] = [stub, REF, ROOT]

from transformers import Qwen2Config, get_scheduler  # noqa: E402
from slamkit.model.unit_lm import UnitLM, UnitLMConfig  # noqa: E402
from oracle import slam_oracle as O  # noqa: E402
from tests.traj_stream import STEPS, LR, WARMUP, MIN_LR, CLIP, stream  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
cfg = O.TINY


def build(dtype):
    base = Qwen2Config(vocab_size=151936, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                       num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                       num_key_value_heads=cfg.n_kv_heads, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                       tie_word_embeddings=True, max_position_embeddings=32768, pad_token_id=0, bos_token_id=1,
                       eos_token_id=1, attention_dropout=0.0)
    m = UnitLM(UnitLMConfig(base_model_name="local", base_config=base, vocab_size=cfg.vocab, twist_init=False))
    sd = O.init_weights(cfg, seed=11, bias_std=0.0, norm_jitter=0.0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("lm_head" in k for k in missing)
    assert m.lm.lm_head.weight.data_ptr() == m.lm.model.embed_tokens.weight.data_ptr()
    return m.to(dtype).train()


def run(bf16):
    m = build(torch.bfloat16 if bf16 else torch.float32)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=LR, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, fused=bool(bf16))
    sched = get_scheduler("cosine_with_min_lr", opt, num_warmup_steps=WARMUP, num_training_steps=STEPS,
                          scheduler_specific_kwargs={"min_lr": MIN_LR})
    losses, lrs, norms = [], [], []
    for step, mb in enumerate(stream()):
        n_items = int((mb["labels"] != -100).sum())   # Trainer.get_batch_samples: labels != -100, unshifted
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if bf16 else torch.autocast("cpu", enabled=False)
        with ctx:
            out = m(input_ids=mb["input_ids"], attention_mask=mb["attention_mask"], labels=mb["labels"],
                    num_items_in_batch=n_items)
        out.loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(params, CLIP)
        lrs.append(sched.get_last_lr()[0])
        opt.step()
        sched.step()
        opt.zero_grad()
        losses.append(float(out.loss))
        norms.append(float(gn))
        if step % 40 == 0 or step == STEPS - 1:
            print(f"{'bf16' if bf16 else 'fp32'} step {step}: loss {losses[-1]:.4f} grad_norm {norms[-1]:.4f} lr {lrs[-1]:.3e}", flush=True)
    final = {k: float(v.detach().float().norm()) for k, v in m.state_dict().items() if "lm_head" not in k}
    return np.array(losses, np.float64), np.array(lrs, np.float64), np.array(norms, np.float64), final


l32, lr32, gn32, f32 = run(False)
l16, lr16, gn16, f16 = run(True)
keys = sorted(f32)
assert keys == sorted(f16)
np.savez_compressed(os.path.join(HERE, "traj.npz"), loss_fp32=l32, lr=lr32, grad_norm_fp32=gn32, loss_bf16=l16, grad_norm_bf16=gn16,
                    final_keys=np.array(keys), final_norm_fp32=np.array([f32[k] for k in keys]),
                    final_norm_bf16=np.array([f16[k] for k in keys]),
                    meta=np.array([STEPS, LR, WARMUP, MIN_LR, CLIP], np.float64))
dev = np.abs(l16 - l32) / l32
print(f"bf16 leg vs fp32 leg: worst single-step deviation {dev.max():.4f} (step {dev.argmax()}), first 60 steps {dev[:60].max():.4f}")
