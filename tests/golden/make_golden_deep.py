"""Depth anchor for the calibrated bf16 tolerances (round 3): the REAL reference (/root/reference slamkit.model.UnitLM over
a local Qwen2Config) on a 12-layer model, once in fp32 and once in its own training precision (bf16 parameters under bf16
autocast, config/model/slam.yaml:9 + training_args bf16). Run in the authoring container:
    HF_HUB_OFFLINE=1 python tests/golden/make_golden_deep.py
Writes tests/golden/deep_model.npz: inputs, fp32 logits / loss / per-tensor gradient norms and samples, and - per decoder
layer - how far the bf16 run's hidden states sit from the fp32 run's, the same for the logits, the loss, and the cosine of
every gradient tensor between the two runs. tests/test_oracle_golden.py holds the oracle to the fp32 numbers and its
`bf16_acts` emulation to the bf16 deviations; the GPU tests state their deep-model bars in units of that emulation."""
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

from transformers import Qwen2Config  # noqa: E402
from slamkit.model.unit_lm import UnitLM, UnitLMConfig  # noqa: E402
from oracle import slam_oracle as O  # noqa: E402

torch.manual_seed(0)
cfg = O.OracleConfig(n_layers=12, hidden=256, n_heads=4, n_kv_heads=2, head_dim=64, intermediate=768)
SEED, BIAS_STD, JIT = 7, 0.02, 0.1


def build():
    base = Qwen2Config(vocab_size=151936, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                       num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                       num_key_value_heads=cfg.n_kv_heads, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                       tie_word_embeddings=True, max_position_embeddings=32768, pad_token_id=0, bos_token_id=1,
                       eos_token_id=1, attention_dropout=0.0)
    m = UnitLM(UnitLMConfig(base_model_name="local", base_config=base, vocab_size=cfg.vocab, twist_init=False))
    sd = O.init_weights(cfg, seed=SEED, bias_std=BIAS_STD, norm_jitter=JIT)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("lm_head" in k for k in missing)
    assert m.lm.lm_head.weight.data_ptr() == m.lm.model.embed_tokens.weight.data_ptr()
    return m.float().train()


g = torch.Generator().manual_seed(5)
lens = [160, 117]
ids = torch.zeros(2, 160, dtype=torch.long)
am = torch.zeros(2, 160, dtype=torch.long)
for b, n in enumerate(lens):
    ids[b, 0] = 1
    ids[b, 1:n] = torch.randint(2, 502, (n - 1,), generator=g)
    am[b, :n] = 1
labels = ids.clone()
labels[am == 0] = -100
valid = am.bool()


def run(m, autocast):
    m.zero_grad()
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else torch.autocast("cpu", enabled=False)
    with ctx:
        inner = m.lm.model(input_ids=ids, attention_mask=am, output_hidden_states=True)
        o = m(input_ids=ids, attention_mask=am, labels=labels)
    o.loss.backward()
    hs = [h.detach().float() for h in inner.hidden_states]  # embeddings, after layer 1 .. L-1, final norm output (HF layout)
    grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
    return o.logits.detach().float(), float(o.loss), hs, grads


def relrms(a, b, mask=None):
    if mask is not None:
        a, b = a[mask], b[mask]
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


m = build()
lg32, loss32, hs32, gr32 = run(m, False)
mb = build().to(torch.bfloat16).train()
lg16, loss16, hs16, gr16 = run(mb, True)

out = {"ids": ids.numpy(), "mask": am.numpy(), "labels": labels.numpy(), "logits_fp32": lg32.numpy().astype(np.float32),
       "loss_fp32": np.float32(loss32), "loss_bf16": np.float32(loss16),
       "logits_bf16_relrms": np.float32(relrms(lg16, lg32, valid)),
       "hidden_bf16_relrms": np.array([relrms(a, b, valid) for a, b in zip(hs16, hs32)], dtype=np.float32),
       "hidden_fp32_rms": np.array([float(h[valid].pow(2).mean().sqrt()) for h in hs32], dtype=np.float32)}
names = sorted(gr32)
out["grad_names"] = np.array(names)
out["grad_norm_fp32"] = np.array([float(gr32[k].norm()) for k in names], dtype=np.float32)
out["grad_cos_bf16_vs_fp32"] = np.array(
    [float((gr16[k].flatten().double() @ gr32[k].flatten().double()) / (gr16[k].norm().double() * gr32[k].norm().double() + 1e-30)) for k in names],
    dtype=np.float64)
for k in names:
    flat = gr32[k].flatten()
    out["gradsample/" + k] = flat[torch.linspace(0, flat.numel() - 1, 16).long()].numpy()
out["meta"] = np.array([repr({"config": cfg.to_dict(), "seed": SEED, "bias_std": BIAS_STD, "norm_jitter": JIT,
                              "transformers": __import__("transformers").__version__, "torch": torch.__version__})])
np.savez_compressed(os.path.join(HERE, "deep_model.npz"), **out)
mat = [c for k, c in zip(names, out["grad_cos_bf16_vs_fp32"]) if k.endswith("proj.weight") or "embed" in k]
print("loss fp32", loss32, "bf16", loss16, "logits relrms", out["logits_bf16_relrms"])
print("hidden relrms per depth", np.round(out["hidden_bf16_relrms"], 5))
print("worst matrix-gradient cosine", min(mat), "worst vector cosine", min(out["grad_cos_bf16_vs_fp32"]))
