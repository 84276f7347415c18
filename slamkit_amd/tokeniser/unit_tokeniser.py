"""UnitTokeniser without the `tokenizers` dependency: same vocabulary, template and methods as
/root/reference slamkit/tokeniser/unit_tokeniser.py:17-120 (pinned by example_data known answers,
tests/test_data_pipeline.py). The feature extractor (HuBERT + k-means) stays the reference's
PyTorch-ROCm path (north_star) and is injected as `speech_tokeniser`.
"""
from __future__ import annotations

import json
import re
from itertools import groupby
from typing import Dict, List, Optional, Sequence, Union

import torch

from .audio_tokeniser import AudioTokeniser


class BatchEncoding(dict):
    """Minimal stand-in for transformers.BatchEncoding: dict with attribute access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class WordLevelUnitVocab:
    """The PreTrainedTokenizerFast(WordLevel) the reference builds (unit_tokeniser.py:33-47):
    '<PAD>'=pad, '<S>'=bos=eos, '<Un i>' = i + offset; split on '>' (merged with previous);
    post-processor '<S> $0 <S>'; padding with pad id."""

    def __init__(self, num_units=500, pad_token_id=0, bos_eos_token_id=1):
        offset = max(bos_eos_token_id, pad_token_id) + 1
        self.vocab: Dict[str, int] = {f"<Un{i}>": i + offset for i in range(num_units)}
        self.vocab.update({"<PAD>": pad_token_id, "<S>": bos_eos_token_id})
        self.inv = {v: k for k, v in self.vocab.items()}
        self.pad_token_id = pad_token_id
        self.bos_token_id = self.eos_token_id = bos_eos_token_id
        self.pad_token, self.bos_token, self.eos_token = "<PAD>", "<S>", "<S>"
        self._split = re.compile(r"[^>]*>")

    def __len__(self):
        return len(self.vocab)

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids = []
        for t in self._split.findall(text.replace(" ", "")):
            if t not in self.vocab:
                raise KeyError(f"unknown unit token {t!r}")
            ids.append(self.vocab[t])
        return [self.bos_token_id] + ids + [self.eos_token_id] if add_special_tokens else ids

    def __call__(self, text: Union[str, Sequence[str]], add_special_tokens=True, return_tensors=None, padding=False,
                 **_) -> BatchEncoding:
        if isinstance(text, str):
            ids = self.encode(text, add_special_tokens)
            enc = BatchEncoding(input_ids=ids, attention_mask=[1] * len(ids))
            if return_tensors == "pt":
                enc = BatchEncoding({k: torch.tensor([v]) for k, v in enc.items()})
            return enc
        rows = [self.encode(t, add_special_tokens) for t in text]
        if padding or return_tensors == "pt":
            n = max(len(r) for r in rows)
            am = [[1] * len(r) + [0] * (n - len(r)) for r in rows]
            rows = [r + [self.pad_token_id] * (n - len(r)) for r in rows]
        else:
            am = [[1] * len(r) for r in rows]
        enc = BatchEncoding(input_ids=rows, attention_mask=am)
        if return_tensors == "pt":
            enc = BatchEncoding({k: torch.tensor(v) for k, v in enc.items()})
        return enc

    def decode(self, ids) -> str:
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        return " ".join(self.inv[int(i)] for i in ids)


class UnitTokeniser(AudioTokeniser):
    def __init__(self, speech_tokeniser=None, dedup: bool = True, bos_eos_token_id: int = 1, pad_token_id: int = 0,
                 num_units: int = 500, load_fe: bool = True):
        super().__init__()
        self.model = speech_tokeniser if load_fe else None
        self.dedup = dedup
        self.bos_token_id = self.eos_token_id = bos_eos_token_id
        self.pad_token_id = pad_token_id
        self.num_units = num_units
        self.text_tokeniser = WordLevelUnitVocab(num_units, pad_token_id, bos_eos_token_id)

    def __call__(self, sample: Union[Dict, str], **kw) -> BatchEncoding:
        if isinstance(sample, dict):
            sample = self.stringify_representation([sample])[0]
        return self.text_tokeniser(sample, **kw)

    def audio_represent(self, wav: torch.Tensor, lens: Optional[torch.Tensor] = None) -> List[Dict]:
        if self.model is None:
            raise RuntimeError("This tokeniser does not have a feature extractor")
        toks = self.model.extract(wav, lens)
        out = []
        for t in toks:
            t = t.tolist()
            if self.dedup:  # unit_tokeniser.py:56-57
                u, d = zip(*[(k, len(list(g))) for k, g in groupby(t)])
            else:
                u, d = t, [1] * len(t)
            out.append({"units": list(u), "duration": list(d)})
        return out

    def stringify_representation(self, reps: List[Dict], mode: str = "test") -> List[str]:
        return ["".join(f"<Un{u}>" for u in cur["units"]) for cur in reps]

    def audio_stringify(self, wav, lens=None) -> List[str]:
        return self.stringify_representation(self.audio_represent(wav, lens))

    def string_tokenise(self, audio_repr, **kw) -> BatchEncoding:
        return self.text_tokeniser(audio_repr, **kw)

    def tokenise(self, wav, lens=None) -> BatchEncoding:
        return self.string_tokenise(self.audio_stringify(wav, lens), return_tensors="pt", padding=True)

    def build_prompt(self, wav, lens=None, output_modality=None) -> BatchEncoding:
        tokens = self.string_tokenise(self.audio_stringify(wav, lens), return_tensors="pt", padding=True)
        return BatchEncoding({k: v[..., :-1] for k, v in tokens.items() if k != "token_type_ids"})

    def prepare_sample(self, sample: dict, **kw) -> BatchEncoding:
        return self.string_tokenise(sample["audio_repr"], **kw)

    def decode_sample(self, tokens: torch.Tensor, output_modality: str = "SPEECH") -> torch.Tensor:
        tokens = tokens[(tokens != self.pad_token_id) & (tokens != self.bos_token_id) & (tokens != self.eos_token_id)]
        offset = max(self.eos_token_id, self.bos_token_id, self.pad_token_id) + 1
        return tokens - offset

    @property
    def fe_sample_rate(self) -> int:
        if self.model is None:
            raise RuntimeError("This tokeniser does not have a feature extractor")
        return self.model.sample_rate

    def save_pretrained(self, save_directory: str, **kw):
        with open(f"{save_directory}/tokeniser_config.json", "w") as f:
            json.dump({"dedup": self.dedup, "bos_eos_token_id": self.bos_token_id, "pad_token_id": self.pad_token_id,
                       "num_units": self.num_units, "load_fe": False}, f)

    @classmethod
    def from_pretrained(cls, path: str, **kw) -> "UnitTokeniser":
        with open(f"{path}/tokeniser_config.json") as f:
            config = json.load(f)
        return cls(speech_tokeniser=None, **config, **kw)

    def get_ignore_tokens(self, _=None):
        return None
