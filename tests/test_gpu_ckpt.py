"""-m gpu: checkpoints written by the REFERENCE load into the engine and reproduce the reference's logits
(fixtures: tests/golden/make_golden_ckpt.py - reference `UnitLM.save_pretrained`, a local HF Qwen2 text LM, and the
reference's TWIST-initialised models on it; /root/reference slamkit/model/unit_lm.py:94-102,200-212).
Tolerance: the engine computes in bf16 from bf16-rounded weights against the reference's fp32 run: logits rel-RMS
<= 2e-2, loss <= 2e-2 abs (the bar of tests/test_gpu_model.py)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN
from tests.gpu_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(GOLDEN, "ckpt_golden.npz")))


def _run(m, gold):
    ids, labels = torch.from_numpy(gold["ids"]), torch.from_numpy(gold["labels"])
    out = m(input_ids=ids, labels=labels)
    torch.cuda.synchronize()
    return out.logits.float().cpu(), float(out.loss)


def test_reference_written_checkpoint_loads(gold):
    from slamkit_amd.model import UnitLM
    m = UnitLM.from_pretrained(os.path.join(GOLDEN, "ref_ckpt"), max_tokens=128)
    assert m.config.vocab_size == 502 and m.config.base_config["rope_theta"] == 10000.0
    logits, loss = _run(m, gold)
    r = rel_err(logits, torch.from_numpy(gold["ref_ckpt_logits"]))
    print(f"[parity] reference checkpoint: logits rel-rms {r:.2e}, loss {loss:.5f} vs {float(gold['ref_ckpt_loss']):.5f}")
    assert r <= 2e-2 and abs(loss - float(gold["ref_ckpt_loss"])) <= 2e-2


@pytest.mark.parametrize("vocab,tag", [(502, "twist502"), (700, "twist700")])
def test_twist_init_from_local_text_lm(gold, vocab, tag):
    from slamkit_amd.model import UnitLM, UnitLMConfig
    path = os.path.join(GOLDEN, "hf_text_lm")
    m = UnitLM(UnitLMConfig(base_model_name=path, twist_init=True, vocab_size=vocab, max_tokens=128))
    ref = torch.from_numpy(gold[f"{tag}_logits"])
    logits, loss = _run(m, gold)
    r = rel_err(logits[..., : ref.shape[-1]], ref)
    print(f"[parity] TWIST vocab {vocab}: logits rel-rms {r:.2e}, loss {loss:.5f} vs {float(gold[tag + '_loss']):.5f}")
    assert r <= 2e-2 and abs(loss - float(gold[f"{tag}_loss"])) <= 2e-2
    if vocab > 640:  # grown rows = mean of the text LM's rows (HF mean-resizing, within its 1e-9-covariance noise)
        from safetensors.torch import load_file
        old = load_file(os.path.join(path, "model.safetensors"))["model.embed_tokens.weight"]
        new = dict(m.named_parameters())["lm.model.embed_tokens.weight"][640:].float().cpu()
        assert float((new - old.mean(0)).abs().max()) <= 1e-3
    # the same weights through from_pretrained on the raw HF directory (un-prefixed `model.*` keys)
    m2 = UnitLM.from_pretrained(path, vocab_size=vocab, max_tokens=128)
    logits2, _ = _run(m2, gold)
    assert torch.equal(logits, logits2)


def test_from_pretrained_rejects_a_mismatching_layout(tmp_path):
    """A checkpoint whose keys do not match must raise, not leave the model randomly initialised."""
    from safetensors.torch import load_file, save_file
    from slamkit_amd.model import UnitLM
    src = os.path.join(GOLDEN, "ref_ckpt")
    sd = load_file(os.path.join(src, "model.safetensors"))
    bad = {k.replace("lm.model.", "decoder."): v for k, v in sd.items()}
    save_file(bad, str(tmp_path / "model.safetensors"))
    with open(os.path.join(src, "config.json")) as f, open(tmp_path / "config.json", "w") as g:
        g.write(f.read())
    with pytest.raises(KeyError, match="missing"):
        UnitLM.from_pretrained(str(tmp_path), max_tokens=128)
    with open(tmp_path / "config.json", "w") as g:
        json.dump({"model_type": "opt", "vocab_size": 502}, g)
    with pytest.raises(ValueError, match="neither"):
        UnitLM.from_pretrained(str(tmp_path), max_tokens=128)
    with pytest.raises(ValueError, match="local HuggingFace checkpoint directory"):
        from slamkit_amd.model import UnitLMConfig
        UnitLMConfig(base_model_name="Qwen/Qwen2.5-0.5B", twist_init=True)
