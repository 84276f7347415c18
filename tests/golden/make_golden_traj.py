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
Writes tests/golden/traj.npz: per-step loss, learning rate and pre-clip gradient norm of both legs, the L2 norm of
every final parameter tensor, and (round 5) `real_bf16` / `real_amp` [8][200]: eight reference-side realisations of the bf16
leg and of HF mixed precision (SDPA / eager attention, fused / unfused AdamW, reversed row order, oneDNN on / off). tests/test_oracle_golden.py holds the oracle's step restatement to the fp32 leg and its
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


def build(dtype, attn="sdpa"):
    base = Qwen2Config(vocab_size=151936, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                       num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                       num_key_value_heads=cfg.n_kv_heads, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                       tie_word_embeddings=True, max_position_embeddings=32768, pad_token_id=0, bos_token_id=1,
                       eos_token_id=1, attention_dropout=0.0)
    base._attn_implementation = attn
    m = UnitLM(UnitLMConfig(base_model_name="local", base_config=base, vocab_size=cfg.vocab, twist_init=False))
    assert m.lm.config._attn_implementation == attn
    sd = O.init_weights(cfg, seed=11, bias_std=0.0, norm_jitter=0.0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("lm_head" in k for k in missing)
    assert m.lm.lm_head.weight.data_ptr() == m.lm.model.embed_tokens.weight.data_ptr()
    return m.to(dtype).train()


def run(leg, attn="sdpa", fused=True, rowperm=False, mkldnn=True, quiet=False):
    """leg: "fp32" (fp32 parameters, no autocast), "bf16" (the recipe: bf16 parameters + autocast + bf16 AdamW state) or "amp"
    (HF mixed precision: fp32 parameters and AdamW state, bf16 autocast - what the engine's fp32-master mode corresponds to).
    attn / fused / rowperm / mkldnn: the SAME run realised differently - attention through SDPA or the eager path
    (modeling_qwen2.py:150-172), torch's fused or for-loop AdamW (transformers 4.48's default optim is the unfused one, 5.x's the
    fused one), the rows of every micro-batch in reversed order (same batch, another summation order), oneDNN on or off."""
    bf16 = leg == "bf16"
    with torch.backends.mkldnn.flags(enabled=mkldnn):
        m = build(torch.bfloat16 if bf16 else torch.float32, attn)
        params = [p for p in m.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=LR, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, fused=bool(fused) if leg != "fp32" else False)
        sched = get_scheduler("cosine_with_min_lr", opt, num_warmup_steps=WARMUP, num_training_steps=STEPS,
                              scheduler_specific_kwargs={"min_lr": MIN_LR})
        losses, lrs, norms = [], [], []
        for step, mb in enumerate(stream()):
            if rowperm:
                mb = {k: v.flip(0) for k, v in mb.items()}
            n_items = int((mb["labels"] != -100).sum())   # Trainer.get_batch_samples: labels != -100, unshifted
            ctx = torch.autocast("cpu", dtype=torch.bfloat16) if leg != "fp32" else torch.autocast("cpu", enabled=False)
            with ctx:
                out = m(input_ids=mb["input_ids"], attention_mask=mb["attention_mask"], labels=mb["labels"],
                        num_items_in_batch=n_items)
            out.loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(params, CLIP)
            lrs.append(sched.get_last_lr()[0])
            opt.step()
            sched.step()
            opt.zero_grad()
            losses.append(float(out.loss.detach()))
            norms.append(float(gn))
            if not quiet and (step % 40 == 0 or step == STEPS - 1):
                print(f"{leg} step {step}: loss {losses[-1]:.4f} grad_norm {norms[-1]:.4f} lr {lrs[-1]:.3e}", flush=True)
        final = {k: float(v.detach().float().norm()) for k, v in m.state_dict().items() if "lm_head" not in k}
    return np.array(losses, np.float64), np.array(lrs, np.float64), np.array(norms, np.float64), final


l32, lr32, gn32, f32 = run("fp32")
l16, lr16, gn16, f16 = run("bf16")
keys = sorted(f32)
assert keys == sorted(f16)
# ---- round 5: the sensitivity envelope from the REFERENCE side. The same 200 steps realised eight ways per precision leg
# (see run()): how far apart these curves sit from each other is what "equivalent to the HF path" can mean at each horizon;
# tests/test_gpu_train.py holds the engine's curve to 1.0 x that spread, no multipliers.
REAL = [dict(attn="sdpa", fused=True, rowperm=False, mkldnn=True), dict(attn="eager", fused=True, rowperm=False, mkldnn=True),
        dict(attn="sdpa", fused=False, rowperm=False, mkldnn=True), dict(attn="eager", fused=False, rowperm=False, mkldnn=True),
        dict(attn="sdpa", fused=True, rowperm=True, mkldnn=True), dict(attn="eager", fused=True, rowperm=True, mkldnn=True),
        dict(attn="sdpa", fused=True, rowperm=False, mkldnn=False), dict(attn="eager", fused=False, rowperm=True, mkldnn=False)]
names = ["+".join(f"{k}={v}" for k, v in r.items()) for r in REAL]
real = {}
for leg in ("bf16", "amp"):
    curves = []
    for r in REAL:
        c = l16 if (leg == "bf16" and r == REAL[0]) else run(leg, quiet=True, **r)[0]
        curves.append(c)
        print(f"{leg} realisation {names[len(curves) - 1]}: final loss {c[-1]:.4f}, worst step vs the first realisation "
              f"{np.max(np.abs(c - curves[0]) / curves[0]):.4f}", flush=True)
    real[leg] = np.stack(curves)
    assert len({c.tobytes() for c in curves}) >= 5, "fewer than five distinct realisations"
np.savez_compressed(os.path.join(HERE, "traj.npz"), loss_fp32=l32, lr=lr32, grad_norm_fp32=gn32, loss_bf16=l16, grad_norm_bf16=gn16,
                    final_keys=np.array(keys), final_norm_fp32=np.array([f32[k] for k in keys]),
                    final_norm_bf16=np.array([f16[k] for k in keys]),
                    real_bf16=real["bf16"], real_amp=real["amp"], real_names=np.array(names),
                    meta=np.array([STEPS, LR, WARMUP, MIN_LR, CLIP], np.float64))
dev = np.abs(l16 - l32) / l32
print(f"bf16 leg vs fp32 leg: worst single-step deviation {dev.max():.4f} (step {dev.argmax()}), first 60 steps {dev[:60].max():.4f}")
