"""CPU: config / weight-file parsing of the checkpoint fixtures written by the reference and by transformers
(tests/golden/make_golden_ckpt.py). The engine side of the same fixtures is tests/test_gpu_ckpt.py."""
import json
import os

import pytest
import torch

from tests.conftest import GOLDEN
from slamkit_amd.model.unit_lm import UnitLM, UnitLMConfig, base_config_from_hf, read_hf_weights

DIMS = dict(num_hidden_layers=2, hidden_size=128, num_attention_heads=2, num_key_value_heads=1, intermediate_size=256)


def test_reference_config_json_is_understood():
    c = json.load(open(os.path.join(GOLDEN, "ref_ckpt", "config.json")))
    b = base_config_from_hf(c["base_config"])  # transformers 5.x layout: rope_theta under rope_parameters
    assert {k: b[k] for k in DIMS} == DIMS and b["rope_theta"] == 10000.0 and b["tie_word_embeddings"] is True
    # transformers 4.x layout
    c4 = dict(c["base_config"], rope_theta=12345.0)
    c4.pop("rope_parameters")
    assert base_config_from_hf(c4)["rope_theta"] == 12345.0
    with pytest.raises(ValueError):
        base_config_from_hf({"model_type": "gpt_neox"})


def test_local_text_lm_directory_as_base_model_name():
    path = os.path.join(GOLDEN, "hf_text_lm")
    cfg = UnitLMConfig(base_model_name=path, twist_init=True, vocab_size=502, rope_theta=10000)
    assert {k: cfg.base_config[k] for k in DIMS} == DIMS and cfg.base_config["head_dim"] == 64
    with pytest.raises(ValueError, match="local HuggingFace checkpoint directory"):
        UnitLMConfig(base_model_name="Qwen/Qwen2.5-0.5B", twist_init=True)
    sd = read_hf_weights(path)
    assert sd["model.embed_tokens.weight"].shape == (640, 128) and len(sd) == 26
    canon = UnitLM._canonical_keys(sd)
    ref = read_hf_weights(os.path.join(GOLDEN, "ref_ckpt"))
    assert set(canon) == set(ref) and all(k.startswith("lm.model.") for k in ref)
    assert ref["lm.model.embed_tokens.weight"].shape == (502, 128) and ref["lm.model.norm.weight"].dtype == torch.float32
