"""Golden vectors for the "config 4"-shaped path (head_dim 128, vocabulary beyond 512, rope_theta 1e6,
GQA group 2), produced by importing the REAL reference (/root/reference) in the authoring container:
    HF_HUB_OFFLINE=1 python tests/golden/make_golden_wide.py
Writes tests/golden/wide_model.npz (inputs + reference outputs only; the reference never ships).
"""
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

from transformers import Qwen2Config, DataCollatorWithFlattening  # noqa: E402
from slamkit.model.unit_lm import UnitLM, UnitLMConfig, compute_loss as ref_compute_loss  # noqa: E402
from oracle import slam_oracle as O  # noqa: E402

torch.manual_seed(0)
cfg = O.OracleConfig(n_layers=2, hidden=256, n_heads=2, n_kv_heads=1, head_dim=128, intermediate=384, vocab=700,
                     rope_theta=1000000.0)
SEED, BIAS_STD, JIT = 5, 0.02, 0.1

base = Qwen2Config(vocab_size=151936, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                   num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads,
                   head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, tie_word_embeddings=True,
                   max_position_embeddings=32768, pad_token_id=0, bos_token_id=1, eos_token_id=1, attention_dropout=0.0)
m = UnitLM(UnitLMConfig(base_model_name="local", base_config=base, vocab_size=cfg.vocab, twist_init=False))
sd = O.init_weights(cfg, seed=SEED, bias_std=BIAS_STD, norm_jitter=JIT)
missing, unexpected = m.load_state_dict(sd, strict=False)
assert not unexpected and all("lm_head" in k for k in missing), (missing, unexpected)
assert m.lm.lm_head.weight.data_ptr() == m.lm.model.embed_tokens.weight.data_ptr()
assert m.lm.model.layers[0].self_attn.head_dim == 128
m = m.float().train()

out = {}
g = torch.Generator().manual_seed(17)
lens = [70, 45, 9]
T = 70
ids = torch.zeros(3, T, dtype=torch.long)
am = torch.zeros(3, T, dtype=torch.long)
for b, n in enumerate(lens):
    ids[b, 0] = 1
    ids[b, 1:n] = torch.randint(2, cfg.vocab, (n - 1,), generator=g)
    am[b, :n] = 1
ids[0, 5:9] = 650  # a repeated id beyond 512: several rows feed one embedding-gradient row
labels = ids.clone()
labels[am == 0] = -100
out["pad_ids"], out["pad_mask"], out["pad_labels"] = ids.numpy(), am.numpy(), labels.numpy()
m.zero_grad()
o = m(input_ids=ids, attention_mask=am, labels=labels)
o.loss.backward()
out["pad_logits"] = o.logits.detach().numpy().astype(np.float32)
out["pad_loss_mean"] = np.float32(o.loss.item())
for k, p in m.named_parameters():
    if p.grad is None:
        continue
    out["pad_gradnorm/" + k] = np.float32(p.grad.norm().item())
for k in ["lm.model.layers.0.self_attn.q_proj.bias", "lm.model.layers.1.self_attn.k_proj.weight",
          "lm.model.layers.0.self_attn.q_proj.weight", "lm.model.norm.weight", "lm.model.layers.1.mlp.down_proj.weight"]:
    out["pad_gradfull/" + k] = dict(m.named_parameters())[k].grad.numpy()
eg = m.lm.model.embed_tokens.weight.grad
rows = torch.tensor([0, 1, 2, 511, 512, 650, 699])
out["pad_embed_grad_rows"] = rows.numpy()
out["pad_embed_grad"] = eg[rows].numpy()

seqs = [[1] + torch.randint(2, cfg.vocab, (n - 1,), generator=g).tolist() for n in (90, 33, 140)]
flat = DataCollatorWithFlattening(return_tensors="pt")([{"input_ids": s} for s in seqs])
with torch.no_grad():
    stitched = torch.cat([m(input_ids=torch.tensor([s])).logits[0] for s in seqs], 0)[None]
out["pack_ids"], out["pack_pos"], out["pack_labels"] = (flat[k].numpy() for k in ("input_ids", "position_ids", "labels"))
out["pack_logits"] = stitched.numpy().astype(np.float32)
out["pack_loss_mean"] = np.float32(ref_compute_loss(stitched, flat["labels"]).item())
m.eval()
out["ll_mean"] = m.log_likelihood(ids.clone(), True).numpy().astype(np.float32)
# modality-restricted scoring (unit_lm.py:187-188): logits of `ignore_tokens` are -inf before the softmax
present = set(ids.flatten().tolist())
ignore = [t for t in range(100, cfg.vocab) if t not in present][:250]
out["ll_ignore_tokens"] = np.array(ignore, dtype=np.int64)
out["ll_ignore_sum"] = m.log_likelihood(ids.clone(), False, ignore_tokens=ignore).numpy().astype(np.float32)

out["meta_config"] = np.array(list(cfg.to_dict().items()), dtype=object).astype(str)
out["meta_init"] = np.array([SEED, BIAS_STD, JIT], dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "wide_model.npz"), **out)
print("wrote wide_model.npz:", {k: getattr(v, "shape", v) for k, v in out.items() if not k.startswith("pad_grad")})
