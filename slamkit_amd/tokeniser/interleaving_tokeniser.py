"""Interleaved speech-text tokeniser (SpiritLM-style) for the scale-up recipe (BASELINE.json configs[3]):
/root/reference slamkit/tokeniser/interleaving_tokeniser.py:59-309 behind the same class and method names.

Host-side integer / string work only. A training row carries `units`, `duration` (frames per unit) and
`aligned_text` = [(word, start_s, end_s), ...]; every word is assigned to the text or the audio modality
(`random` coin flips, `span` patience runs, or `poisson` span selection), runs of equal modality become either
the words themselves or the units whose time span covers the run, and `<speech>` / `<text>` markers separate
the runs. The string is then tokenised by a TEXT tokenizer extended with `<Un0>..<Un{n-1}>, <speech>, <text>`.

The text tokenizer is a `transformers` tokenizer loaded from `text_tokeniser_path` (a local directory on a
training box: the hub is not reachable); only this class needs `transformers`. The HuBERT feature extractor
stays the reference's path (north_star): it is injected as `speech_tokeniser` when audio has to be encoded,
and only its `get_unit_duration()` matters for the prepare_tokens / train stages, so `unit_duration` may be
given directly instead. Bit-exact parity with the reference on seeded RNG streams is pinned by
tests/golden/interleave.json (tests/test_data_pipeline.py): the draws from `torch.rand` / `np.random` happen
in the same order as in the reference.
"""
from __future__ import annotations

import math
import re
from bisect import bisect_left, bisect_right
from itertools import groupby
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .audio_tokeniser import AudioTokeniser

SPEECH_TOKEN = "<speech>"
TEXT_TOKEN = "<text>"


def select_spans_poisson(array_size: int, lambda_param: int, eta: float) -> np.ndarray:
    """0/1 mask over `array_size` words with about ceil(eta * size) ones in non-overlapping spans whose lengths
    are Poisson(lambda) (interleaving_tokeniser.py:57-96). Start positions are drawn uniformly from the
    positions that are neither selected nor directly behind a selected span."""
    want = math.ceil(array_size * eta)
    mask = np.zeros(array_size, dtype=int)
    free = set(range(array_size))
    got = 0
    while got < want and free:
        first = np.random.choice(list(free))
        last = min(first + np.random.poisson(lambda_param), array_size)  # exclusive
        if mask[first:last].any():
            continue  # would run into an earlier span: draw again
        mask[first:last] = 1
        got += last - first
        free.difference_update(range(first, last))
        if last < array_size:
            free.discard(last)
    return mask


class InterleavingTokeniser(AudioTokeniser):
    def __init__(self, speech_tokeniser=None, dedup: bool = True, pad_token_id: int = 0, num_units: int = 500,
                 load_fe: bool = True, text_tokeniser_path: str = "facebook/opt-125m", interleave_method: str = "random",
                 interleave_span: Optional[int] = None, interleave_prob: Optional[float] = None,
                 unit_duration: Optional[float] = None):
        super().__init__()
        self.speech_fe = speech_tokeniser if load_fe else None
        self.dedup = dedup
        self.pad_token_id = pad_token_id
        self.num_units = num_units
        self.text_tokeniser = self._init_text_tokeniser(text_tokeniser_path, pad_token_id, num_units)
        self.interleave_method = interleave_method
        self.interleave_span = interleave_span
        self.interleave_prob = interleave_prob
        self._unit_duration = unit_duration

    @staticmethod
    def _init_text_tokeniser(path: str, pad_token_id: int, num_units: int):
        """interleaving_tokeniser.py:122-127: the text tokenizer + one token per unit + the two markers."""
        try:
            from transformers import AutoTokenizer
        except ImportError as e:  # pragma: no cover
            raise RuntimeError("InterleavingTokeniser needs `transformers` for the text side") from e
        tok = AutoTokenizer.from_pretrained(path)
        tok.pad_token_id = pad_token_id
        tok.padding_side = "right"
        tok.add_tokens([f"<Un{u}>" for u in range(num_units)] + [SPEECH_TOKEN, TEXT_TOKEN])
        return tok

    # ---- unit timing ------------------------------------------------------------------------------------
    def unit_duration(self) -> float:
        if self._unit_duration is not None:
            return float(self._unit_duration)
        if self.speech_fe is None:
            raise RuntimeError("interleaving needs the duration of one unit: pass unit_duration= or a feature extractor")
        return float(self.speech_fe.get_unit_duration())

    # ---- audio -> units (needs the injected feature extractor) ---------------------------------------
    @torch.inference_mode()
    def audio_represent(self, wav, lens=None) -> List[Dict]:
        if self.speech_fe is None:
            raise RuntimeError("This tokeniser does not have a feature extractor")
        out = []
        for t in self.speech_fe.extract(wav, lens):
            t = t.tolist()
            if self.dedup:
                u, d = zip(*[(k, sum(1 for _ in grp)) for k, grp in groupby(t)])
            else:
                u, d = t, [1] * len(t)
            out.append({"units": u, "duration": d})
        return out

    # ---- modality assignment (interleaving_tokeniser.py:138-161) --------------------------------------
    def _assign_interleaved_modality(self, aligned_text: Sequence) -> List[Tuple]:
        words = [tuple(w) for w in aligned_text]
        if self.interleave_method == "random":
            flags = [not bool(torch.rand(1) < 0.5) for _ in words]  # True = audio
        elif self.interleave_method == "span":
            flags, run = [], 0
            for _ in words:
                if not bool(torch.rand(1) >= self.interleave_prob):  # this word opens (or extends) an audio run
                    run = self.interleave_span
                flags.append(run > 0)
                run -= 1
        elif self.interleave_method == "poisson":
            sel = select_spans_poisson(len(words), self.interleave_span, self.interleave_prob)
            flags = [bool(s > 0) for s in sel]
        else:
            flags = []  # the reference yields no modalities (and an empty string) for an unknown method
            words = []
        return [(w, s, e, "audio" if a else "text") for (w, s, e), a in zip(words, flags)]

    def _create_interleaved_text(self, rep: Dict, aligned_text: Sequence[Tuple]) -> str:
        if not aligned_text:
            return ""
        ends = np.cumsum(rep["duration"]) * self.unit_duration()  # end time of every unit
        runs = [(m, list(grp)) for m, grp in groupby(aligned_text, key=lambda x: x[3])]
        pieces = [TEXT_TOKEN if runs[0][0] == "text" else SPEECH_TOKEN]
        for i, (m, grp) in enumerate(runs):
            if m == "text":
                pieces.append("".join(w for w, _, _, _ in grp))
            else:
                lo, hi = bisect_left(ends, grp[0][1]), bisect_right(ends, grp[-1][2])
                pieces.append("".join(f"<Un{u}>" for u in rep["units"][lo:hi]))
            if i + 1 < len(runs):  # marker of the modality that follows
                pieces.append(SPEECH_TOKEN if m == "text" else TEXT_TOKEN)
        return "".join(pieces)

    def _interleave_units(self, rep: Dict) -> str:
        return self._create_interleaved_text(rep, self._assign_interleaved_modality(rep["aligned_text"]))

    def stringify_representation(self, reps: List[Dict], mode: str = "test") -> List[str]:
        out = []
        for cur in reps:
            if mode == "train":
                out.append(self._interleave_units(cur))
            elif mode == "test":
                out.append("".join(f"<Un{u}>" for u in cur["units"]))
        return out

    # ---- strings -> ids -----------------------------------------------------------------------------------
    def string_tokenise(self, audio_repr, **kw):
        return self.text_tokeniser(audio_repr, add_special_tokens=True, **kw)

    def prepare_sample(self, sample: dict, **kw):
        return self.string_tokenise(sample["audio_repr"], **kw)

    def _stringify_interleaved(self, inp) -> str:
        """[("TEXT" | "SPEECH", content), ...] (or objects with .content_type / .content) -> one string; speech
        content is a waveform tensor and needs the injected feature extractor."""
        parts, prev = [], None
        for seg in inp:
            kind, content = seg if isinstance(seg, tuple) else (seg.content_type, seg.content)
            kind = str(getattr(kind, "value", kind)).upper()
            if kind == "SPEECH":
                if prev != "s":
                    parts.append(SPEECH_TOKEN)
                parts.append(self.stringify_representation(self.audio_represent(content.unsqueeze(0)))[0])
                prev = "s"
            elif kind == "TEXT":
                if prev != "t":
                    parts.append(TEXT_TOKEN)
                parts.append(content)
                prev = "t"
            else:
                raise ValueError(f"Unknown content type: {kind}")
        return "".join(parts)

    def tokenise(self, inputs, lens=None):
        if isinstance(inputs, torch.Tensor):
            strs = self.stringify_representation(self.audio_represent(inputs, lens))
        elif isinstance(inputs, list):
            strs = [self._stringify_interleaved(i) for i in inputs]
        else:
            raise ValueError(f"Inputs should be a list of interleaved inputs or a torch.Tensor, got {type(inputs)}")
        return self.string_tokenise(strs, return_tensors="pt", padding=True)

    def build_prompt(self, inputs, lens=None, output_modality=None):
        if isinstance(inputs, torch.Tensor):
            strs = self.stringify_representation(self.audio_represent(inputs, lens))
        elif isinstance(inputs, list):
            strs = [self._stringify_interleaved(i) for i in inputs]
        else:
            raise ValueError(f"Inputs should be a list of interleaved inputs or a torch.Tensor, got {type(inputs)}")
        if output_modality:
            m = output_modality.upper()
            if m not in ("SPEECH", "TEXT"):
                raise ValueError(f"Unknown output modality: {output_modality}")
            strs = [s + (SPEECH_TOKEN if m == "SPEECH" else TEXT_TOKEN) for s in strs]
        tokens = self.string_tokenise(strs, return_tensors="pt", padding=True)
        eos = self.text_tokeniser.eos_token_id
        if eos is not None and bool((tokens["input_ids"][..., -1] == eos).any()):
            tokens = {k: v[..., :-1] for k, v in tokens.items()}
        return tokens

    # ---- ids -> units / text ---------------------------------------------------------------------------
    def _marker_ids(self) -> List[int]:
        return [self.text_tokeniser.encode(SPEECH_TOKEN)[0], self.text_tokeniser.encode(TEXT_TOKEN)[0]]

    def get_ignore_tokens(self, used_token_modality: Optional[str]) -> Optional[List[int]]:
        """Token ids that do NOT belong to `used_token_modality` (interleaving_tokeniser.py:290-308)."""
        tt = self.text_tokeniser
        n_text = len(tt) - self.num_units - 2
        special = [tt.bos_token_id, tt.eos_token_id]
        if used_token_modality and used_token_modality.upper() == "SPEECH":
            return [x for x in range(n_text) if x not in special] + self._marker_ids()
        if used_token_modality and used_token_modality.upper() == "TEXT":
            skip = special + self._marker_ids()
            return [x for x in range(n_text, len(tt)) if x not in skip]
        return None

    def decode_sample(self, tokens: torch.Tensor, output_modality: str = "SPEECH"):
        tt = self.text_tokeniser
        drop = [i for i in (tt.pad_token_id, tt.bos_token_id, tt.eos_token_id) if i is not None] + self._marker_ids()
        if output_modality:
            drop += self.get_ignore_tokens(output_modality)
        keep = tokens[~torch.isin(tokens, torch.tensor(drop, device=tokens.device))]
        text = tt.decode(keep)
        if output_modality.upper() == "SPEECH":
            return torch.tensor([int(n) for n in re.findall(r"<Un(\d+)>", text)])
        if output_modality.upper() == "TEXT":
            return text
        raise ValueError(f"Unknown output modality: {output_modality}")

    @property
    def fe_sample_rate(self) -> int:
        if self.speech_fe is None:
            raise RuntimeError("This tokeniser does not have a feature extractor")
        return self.speech_fe.sample_rate
