"""Reference-WRITTEN checkpoints as fixtures. Run in the authoring container:
    HF_HUB_OFFLINE=1 python tests/golden/make_golden_ckpt.py
Writes (data only - safetensors weights, json configs, npz inputs/outputs):
  tests/golden/ref_ckpt/            <- reference `UnitLM.save_pretrained` of a tiny random-init model
                                       (slamkit/model/unit_lm.py:82-112; key layout `lm.model.layers.*`)
  tests/golden/hf_text_lm/          <- `transformers.Qwen2ForCausalLM.save_pretrained` of a tiny text LM with a
                                       640-row vocabulary: what `base_model_name` points at for TWIST initialisation
  tests/golden/ckpt_golden.npz      <- token batch + fp32 logits / loss of (a) the reference model reloaded from
                                       ref_ckpt with `UnitLM.from_pretrained`, (b) the reference `UnitLM(config with
                                       twist_init=True, base_model_name=hf_text_lm, vocab_size=502)`
                                       (unit_lm.py:94-102: AutoModelForCausalLM.from_pretrained then
                                       resize_token_embeddings -> first 502 embedding rows), and (c) the same with
                                       vocab_size=700 > 640 rows (the grown rows: HF mean-resizing; only the rows that
                                       existed are pinned, through logits restricted to ids < 640).
"""
import os
import shutil
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

from transformers import Qwen2Config, Qwen2ForCausalLM  # noqa: E402
from slamkit.model.unit_lm import UnitLM, UnitLMConfig  # noqa: E402

# transformers 5.x `PreTrainedConfig.save_pretrained` builds `self.__class__()` to diff against the defaults; the reference's
# default base model is a hub id (unit_lm.py:36 "facebook/opt-350M"), unreachable here. Composite configs declare
# `has_no_defaults_at_init`; set it on the imported class at run time (the reference source is untouched).
UnitLMConfig.has_no_defaults_at_init = True

torch.manual_seed(0)
DIMS = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
            rms_norm_eps=1e-6, rope_theta=10000.0, tie_word_embeddings=True, max_position_embeddings=4096,
            attention_dropout=0.0)
V = 502


def randomise(m, seed):
    """HF init leaves biases at 0 and norms at 1: perturb them so the fixture exercises every tensor."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith(".bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif k.endswith("norm.weight"):
                p.copy_(1.0 + torch.randn(p.shape, generator=g) * 0.1)
    return m


out = {}
g = torch.Generator().manual_seed(7)
ids = torch.randint(2, V, (2, 48), generator=g)
ids[:, 0] = 1
labels = ids.clone()
labels[1, 40:] = -100
out["ids"], out["labels"] = ids.numpy(), labels.numpy()

# ---- the local HF text LM (TWIST source, and the base-config anchor of the reference checkpoint) ------------------------------------------------------------
torch.manual_seed(1)
text = randomise(Qwen2ForCausalLM(Qwen2Config(vocab_size=640, pad_token_id=0, bos_token_id=1, eos_token_id=1, **DIMS)).float(), 12)
t = os.path.join(HERE, "hf_text_lm")
shutil.rmtree(t, ignore_errors=True)
text.save_pretrained(t, safe_serialization=True)
# ---- (a) reference-written UnitLM checkpoint -----------------------------------------------------------------------
# base_model_name = the local text-LM directory: the reference's from_pretrained re-resolves it through
# AutoConfig.from_pretrained(base_model_name, **base_config) (unit_lm.py:68-70), which must not reach for the hub
torch.manual_seed(2)
m = randomise(UnitLM(UnitLMConfig(base_model_name=t, vocab_size=V, twist_init=False, torch_dtype=torch.float32)).float(), 11)
d = os.path.join(HERE, "ref_ckpt")
shutil.rmtree(d, ignore_errors=True)
m.save_pretrained(d, safe_serialization=True)
m2 = UnitLM.from_pretrained(d).float().eval()
with torch.no_grad():
    o = m2(input_ids=ids, labels=labels)
out["ref_ckpt_logits"], out["ref_ckpt_loss"] = o.logits.float().numpy(), np.float32(o.loss)
with torch.no_grad():
    o0 = m.eval()(input_ids=ids, labels=labels)
assert torch.allclose(o0.logits, o.logits, atol=1e-6), "reference save/load is not a round trip"

# ---- (b), (c) TWIST initialisation from the local HF text LM ------------------------------------------------------------
for tag, vocab in (("twist502", 502), ("twist700", 700)):
    torch.manual_seed(5)
    cfg = UnitLMConfig(base_model_name=t, vocab_size=vocab, twist_init=True, torch_dtype=torch.float32)
    tw = UnitLM(cfg).float().eval()
    emb = tw.lm.get_input_embeddings().weight
    assert emb.shape[0] == vocab and torch.equal(emb[: min(vocab, 640)], text.get_input_embeddings().weight[: min(vocab, 640)])
    with torch.no_grad():
        o = tw(input_ids=ids, labels=labels)
    lg = o.logits.float()
    out[f"{tag}_logits"] = lg[..., : min(vocab, 640)].numpy()
    out[f"{tag}_loss"] = np.float32(o.loss)
    if vocab > 640:
        new = emb[640:].detach()
        old = text.get_input_embeddings().weight.detach()
        out["twist700_new_rows_absdev_from_mean"] = np.float32((new - old.mean(0)).abs().max())
np.savez_compressed(os.path.join(HERE, "ckpt_golden.npz"), **out)
for p in (d, t):
    for fn in sorted(os.listdir(p)):
        print(p, fn, os.path.getsize(os.path.join(p, fn)))
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
