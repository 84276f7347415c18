"""Generate golden vectors by importing the REAL reference (/root/reference) in the authoring
container. Run:  HF_HUB_OFFLINE=1 python tests/golden/make_golden.py
Writes tests/golden/{tiny_model.npz,data.json}. The reference itself never ships; only these
inputs/outputs do. Needs a 3-name omegaconf stub (absent package) created on the fly in a tmp dir.
"""
import json
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

from transformers import Qwen2Config, DataCollatorForLanguageModeling, DataCollatorWithFlattening  # noqa: E402
from slamkit.model.unit_lm import UnitLM, UnitLMConfig  # noqa: E402
from slamkit.tokeniser.unit_tokeniser import UnitTokeniser  # noqa: E402
from slamkit.data.hf_dataset import chunk_texts  # noqa: E402
from slamkit.utils.calculation_utils import calc_nll  # noqa: E402
from oracle import slam_oracle as O  # noqa: E402

torch.manual_seed(0)
cfg = O.TINY
SEED, BIAS_STD, JIT = 3, 0.02, 0.1


def build_ref(cfg):
    base = Qwen2Config(vocab_size=151936, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                       num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                       num_key_value_heads=cfg.n_kv_heads, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                       tie_word_embeddings=True, max_position_embeddings=32768, pad_token_id=0, bos_token_id=1,
                       eos_token_id=1, attention_dropout=0.0)
    ucfg = UnitLMConfig(base_model_name="local", base_config=base, vocab_size=cfg.vocab, twist_init=False)
    m = UnitLM(ucfg)
    sd = O.init_weights(cfg, seed=SEED, bias_std=BIAS_STD, norm_jitter=JIT)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("lm_head" in k for k in missing), (missing, unexpected)
    m.lm.tie_weights() if hasattr(m.lm, "tie_weights") else None
    assert m.lm.lm_head.weight.data_ptr() == m.lm.model.embed_tokens.weight.data_ptr()
    return m.float().train(), sd


def grads_of(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


out = {}
data = {}

# ---- G1: tokeniser known answers on the reference's example_data ---------------------------
tok = UnitTokeniser(None, load_fe=False)
rows = [json.loads(l) for l in open(os.path.join(REF, "example_data", "tokens.jsonl"))]
feats = [json.loads(l) for l in open(os.path.join(REF, "example_data", "features.jsonl"))]
g1 = []
for r, f in zip(rows, feats):
    assert tok.stringify_representation([f], mode="train")[0] == r["audio_repr"]
    enc = tok.prepare_sample(r)
    g1.append({"units": f["units"], "audio_repr": r["audio_repr"], "input_ids": list(enc["input_ids"]),
               "attention_mask": list(enc["attention_mask"])})
data["G1_tokens"] = g1
data["vocab_size"] = len(tok.text_tokeniser)

# ---- G2: chunk_texts ------------------------------------------------------------------------
ex = {"input_ids": [g["input_ids"] for g in g1], "attention_mask": [g["attention_mask"] for g in g1]}
data["G2_chunks"] = {str(c): {k: v for k, v in chunk_texts(ex, c).items()} for c in (128, 512, 1024)}

# ---- G3: collators on the 128-chunks ----------------------------------------------------------
ch = chunk_texts(ex, 128)
features = [{"input_ids": a, "attention_mask": b} for a, b in zip(ch["input_ids"], ch["attention_mask"])]
lm = DataCollatorForLanguageModeling(tok.text_tokeniser, mlm=False, return_tensors="pt")(features[1:4])
fl = DataCollatorWithFlattening(return_tensors="pt")([{"input_ids": f["input_ids"]} for f in features[1:4]])
data["G3_lm"] = {k: v.tolist() for k, v in lm.items()}
data["G3_flat"] = {k: v.tolist() for k, v in fl.items() if hasattr(v, "tolist")}

# ---- G4: tiny UnitLM, padded batch ------------------------------------------------------------
m, sd = build_ref(cfg)
g = torch.Generator().manual_seed(11)
lens = [37, 23]
ids = torch.zeros(2, 37, dtype=torch.long)
am = torch.zeros(2, 37, dtype=torch.long)
for b, n in enumerate(lens):
    ids[b, 0] = 1
    ids[b, 1:n] = torch.randint(2, 502, (n - 1,), generator=g)
    am[b, :n] = 1
labels = ids.clone()
labels[am == 0] = -100
out["pad_ids"], out["pad_mask"], out["pad_labels"] = ids.numpy(), am.numpy(), labels.numpy()

m.zero_grad()
o = m(input_ids=ids, attention_mask=am, labels=labels)
o.loss.backward()
out["pad_logits"] = o.logits.detach().numpy().astype(np.float32)
out["pad_loss_mean"] = np.float32(o.loss.item())
gr = grads_of(m)
nitems = int((labels != -100).sum())
m.zero_grad()
o2 = m(input_ids=ids, attention_mask=am, labels=labels, num_items_in_batch=nitems)
out["pad_loss_sum"] = np.float32(o2.loss.item())
out["pad_num_items"] = np.int64(nitems)
for k, v in gr.items():
    out["pad_gradnorm/" + k] = np.float32(v.norm().item())
    flat = v.flatten()
    idx = torch.linspace(0, flat.numel() - 1, 16).long()
    out["pad_gradsample/" + k] = flat[idx].numpy()
# full grads of a few tensors (small ones + embedding)
for k in ["lm.model.embed_tokens.weight", "lm.model.layers.0.self_attn.q_proj.bias",
          "lm.model.layers.1.self_attn.k_proj.bias", "lm.model.layers.0.input_layernorm.weight",
          "lm.model.norm.weight", "lm.model.layers.1.self_attn.v_proj.weight"]:
    out["pad_gradfull/" + k] = gr[k].numpy()

# ---- G4b: packed batch ------------------------------------------------------------------------
seqs = [[1] + torch.randint(2, 502, (n - 1,), generator=g).tolist() for n in (29, 41, 17)]
flat = DataCollatorWithFlattening(return_tensors="pt")([{"input_ids": s} for s in seqs])
pid, ppos, plab = flat["input_ids"], flat["position_ids"], flat["labels"]
# per-sequence reference: run each sequence alone and stitch (what varlen attention computes)
logits_sep = []
with torch.no_grad():
    for s in seqs:
        logits_sep.append(m(input_ids=torch.tensor([s])).logits[0])
stitched = torch.cat(logits_sep, 0)[None]
from slamkit.model.unit_lm import compute_loss as ref_compute_loss  # noqa: E402
out["pack_ids"], out["pack_pos"], out["pack_labels"] = pid.numpy(), ppos.numpy(), plab.numpy()
out["pack_logits"] = stitched.numpy().astype(np.float32)
out["pack_loss_mean"] = np.float32(ref_compute_loss(stitched, plab).item())
# does the reference model itself honour position_ids packing on this transformers version?
with torch.no_grad():
    direct = m(input_ids=pid, position_ids=ppos).logits
out["pack_direct_maxdiff"] = np.float32((direct - stitched).abs().max().item())

# ---- G7: log-likelihood -------------------------------------------------------------------------
m.eval()
toks = ids.clone()
for mean_nll in (True, False):
    ll = m.log_likelihood(toks.clone(), mean_nll)
    out[f"ll_{'mean' if mean_nll else 'sum'}"] = ll.numpy().astype(np.float32)

# ---- G5: bf16 params + bf16 autocast (tolerance calibration only) -------------------------------
mb = m.to(torch.bfloat16).train()
with torch.autocast("cpu", dtype=torch.bfloat16):
    ob = mb(input_ids=ids, attention_mask=am, labels=labels)
out["pad_loss_bf16"] = np.float32(ob.loss.item())
out["pad_logits_bf16_rmsrel"] = np.float32(
    ((ob.logits.float() - torch.from_numpy(out["pad_logits"])).pow(2).mean().sqrt()
     / torch.from_numpy(out["pad_logits"]).pow(2).mean().sqrt()).item())

# ---- G9: SLAMDPOTrainer.tokenize_row (slam_dpo_trainer.py:7-64), loaded by file path over a stub `trl` ----
import importlib.util  # noqa: E402
import types  # noqa: E402
trl_stub = types.ModuleType("trl")
trl_stub.DPOTrainer = type("DPOTrainer", (), {})
sys.modules["trl"] = trl_stub
spec = importlib.util.spec_from_file_location("ref_slam_dpo_trainer", os.path.join(REF, "slamkit", "trainer", "slam_dpo_trainer.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
gg = torch.Generator().manual_seed(21)
g9 = []
for i in range(6):
    mk = lambda n: "".join(f"<Un{int(u)}>" for u in torch.randint(0, 500, (n,), generator=gg))  # noqa: E731
    feats = {"prompt": mk(int(torch.randint(3, 12, (1,), generator=gg))), "chosen": mk(int(torch.randint(2, 9, (1,), generator=gg))),
             "rejected": mk(int(torch.randint(2, 9, (1,), generator=gg)))}
    mp, mc = [(None, None), (5, 4), (8, None), (None, 3), (4, 6), (6, 2)][i]
    row = mod.SLAMDPOTrainer.tokenize_row(feats, tok, mp, mc, False)  # the CLI passes the UnitTokeniser itself (preference_alignment_train.py:56-58)
    g9.append({"features": feats, "max_prompt_length": mp, "max_completion_length": mc, "row": {k: list(v) for k, v in row.items()}})
data["G9_dpo_rows"] = g9

meta = {"config": cfg.to_dict(), "seed": SEED, "bias_std": BIAS_STD, "norm_jitter": JIT,
        "transformers": __import__("transformers").__version__, "torch": torch.__version__}
data["meta"] = meta
np.savez_compressed(os.path.join(HERE, "tiny_model.npz"), **out)
with open(os.path.join(HERE, "data.json"), "w") as f:
    json.dump(data, f)
print("wrote", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith("pad_grad")})
print("pack_direct_maxdiff", out["pack_direct_maxdiff"], "bf16 loss", out["pad_loss_bf16"], "fp32", out["pad_loss_mean"],
      "rmsrel", out["pad_logits_bf16_rmsrel"])
