"""Dataset side of the hot path: jsonl of `audio_repr` strings -> token ids -> context_len chunks ->
collated int64 batches. Same functions and semantics as /root/reference slamkit/data/hf_dataset.py
(split_into_chunks :16-18, chunk_texts :21-26, init_dataset :29-66, get_filter_fn :69-88,
parse_single_dataset :91-118, init_preference_optimization_dataset :138-148) without the HF `datasets`
machinery; plus a binary pre-tokenised shard format that `saved_ds_path` uses.
"""
from __future__ import annotations

import json
import logging
import os
from glob import glob
from itertools import chain
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

logger = logging.getLogger(__name__)


def _get(cfg, key, default=None):
    if cfg is None:
        return default
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


def split_into_chunks(lst, chunk_size):
    return [lst[i:i + chunk_size] for i in range(0, len(lst), chunk_size)]


def chunk_texts(examples: Dict[str, List[List[int]]], chunk_size: int):
    """Consecutive chunk_size slices, remainder kept, no special tokens re-added (hf_dataset.py:21-26)."""
    return {k: list(chain.from_iterable(split_into_chunks(l, chunk_size) for l in v)) for k, v in examples.items()}


def get_filter_fn(sample_units_min_length=None, sample_units_max_length=None):
    assert sample_units_min_length is not None or sample_units_max_length is not None, \
        "At least one of sample_units_min_length or sample_units_max_length should be non None"
    if sample_units_min_length is None:
        return lambda x: len(x["input_ids"]) <= sample_units_max_length
    if sample_units_max_length is None:
        return lambda x: len(x["input_ids"]) >= sample_units_min_length
    return lambda x: sample_units_min_length <= len(x["input_ids"]) <= sample_units_max_length


class TokenDataset:
    """In-memory list of {'input_ids', 'attention_mask'} rows."""

    def __init__(self, rows: List[Dict[str, List[int]]]):
        self.rows = rows

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        return self.rows[i]

    def filter(self, fn):
        return TokenDataset([r for r in self.rows if fn(r)])

    @property
    def num_tokens(self):
        return sum(len(r["input_ids"]) for r in self.rows)


# ---- binary shard format: tokens.bin (uint16 LE, all sequences back to back) + index.npy (int64
#      offsets, n+1 entries) + meta.json. ids = unit + 2 exactly as UnitTokeniser produces them. -----
def write_token_shard(path: str, dataset) -> None:
    os.makedirs(path, exist_ok=True)
    lens = np.fromiter((len(dataset[i]["input_ids"]) for i in range(len(dataset))), dtype=np.int64, count=len(dataset))
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    buf = np.empty(int(off[-1]), dtype="<u2")
    for i in range(len(dataset)):
        ids = dataset[i]["input_ids"]
        if len(ids) and (max(ids) > 65535 or min(ids) < 0):
            raise ValueError("token id does not fit uint16")
        buf[off[i]:off[i + 1]] = ids
    buf.tofile(os.path.join(path, "tokens.bin"))
    np.save(os.path.join(path, "index.npy"), off)
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump({"format": "slam-token-shard-v1", "dtype": "uint16", "sequences": int(len(lens)),
                   "tokens": int(off[-1])}, f)


class TokenShardDataset:
    """Memory-mapped view of a shard written by write_token_shard."""

    def __init__(self, path: str):
        self.off = np.load(os.path.join(path, "index.npy"))
        self.tok = np.memmap(os.path.join(path, "tokens.bin"), dtype="<u2", mode="r")
        assert int(self.off[-1]) == self.tok.shape[0]

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        ids = self.tok[self.off[i]:self.off[i + 1]].astype(np.int64).tolist()
        return {"input_ids": ids, "attention_mask": [1] * len(ids)}

    @property
    def num_tokens(self):
        return int(self.off[-1])


# ---- collators ----------------------------------------------------------------------------------------
class DataCollatorForLanguageModeling:
    """transformers.DataCollatorForLanguageModeling(mlm=False): right-pad to the longest row,
    labels = input_ids with pad -> -100 (hf_dataset.py:64)."""

    def __init__(self, tokenizer=None, mlm: bool = False, return_tensors: str = "pt", pad_token_id: Optional[int] = None):
        assert not mlm
        self.pad_id = pad_token_id if pad_token_id is not None else getattr(tokenizer, "pad_token_id", 0)

    def __call__(self, features: Sequence[Dict[str, List[int]]]) -> Dict[str, torch.Tensor]:
        T = max(len(f["input_ids"]) for f in features)
        ids = torch.full((len(features), T), self.pad_id, dtype=torch.long)
        am = torch.zeros((len(features), T), dtype=torch.long)
        for i, f in enumerate(features):
            n = len(f["input_ids"])
            ids[i, :n] = torch.as_tensor(f["input_ids"], dtype=torch.long)
            am[i, :n] = 1
        labels = ids.clone()
        labels[labels == self.pad_id] = -100
        return {"input_ids": ids, "attention_mask": am, "labels": labels}


class DataCollatorWithFlattening:
    """transformers.DataCollatorWithFlattening: one [1, sum T] row, position_ids restarting per
    sequence, labels[first token of each sequence] = -100 (hf_dataset.py:61-62)."""

    def __init__(self, return_tensors: str = "pt"):
        pass

    def __call__(self, features: Sequence[Dict[str, List[int]]]) -> Dict[str, torch.Tensor]:
        ids, pos, lab = [], [], []
        for f in features:
            x = list(f["input_ids"])
            ids += x
            pos += list(range(len(x)))
            lab += [-100] + x[1:]
        return {"input_ids": torch.tensor([ids]), "position_ids": torch.tensor([pos]), "labels": torch.tensor([lab])}


# ---- dataset construction --------------------------------------------------------------------------------
def _read_jsonl(paths: Sequence[str]):
    for p in sorted(paths):
        with open(p) as f:
            for line in f:
                line = line.strip()
                if line:
                    yield json.loads(line)


def _build_split(cfg, tokeniser, pattern: Optional[str], is_train: bool) -> Optional[TokenDataset]:
    if pattern is None:
        return None
    rows = []
    for r in _read_jsonl(glob(pattern)):
        enc = tokeniser.prepare_sample(r)
        rows.append({"input_ids": list(enc["input_ids"]), "attention_mask": list(enc["attention_mask"])})
    ds = TokenDataset(rows)
    data = _get(cfg, "data")
    if is_train and _get(data, "sample_units_max_length", None):
        ds = ds.filter(get_filter_fn(sample_units_max_length=_get(data, "sample_units_max_length")))
    ctx = _get(_get(cfg, "model"), "context_len", None)
    if ctx is not None:
        ch = chunk_texts({"input_ids": [r["input_ids"] for r in ds.rows],
                          "attention_mask": [r["attention_mask"] for r in ds.rows]}, ctx)
        ds = TokenDataset([{"input_ids": a, "attention_mask": b} for a, b in zip(ch["input_ids"], ch["attention_mask"])])
    if is_train and _get(data, "chunk_units_min_length", None):
        ds = ds.filter(get_filter_fn(sample_units_min_length=_get(data, "chunk_units_min_length")))
    return ds


def parse_single_dataset(cfg, tokeniser, train_path: str, val_path: Optional[str] = None) -> Dict[str, TokenDataset]:
    out = {"train": _build_split(cfg, tokeniser, train_path, True)}
    v = _build_split(cfg, tokeniser, val_path, False)
    if v is not None:
        out["validation"] = v
    return out


def interleave_datasets(datasets: List[TokenDataset], probabilities: List[float], seed: int = 0,
                        stopping_strategy: str = "first_exhausted") -> TokenDataset:
    """Seeded probabilistic interleave (datasets.interleave_datasets semantics, seed 0 at hf_dataset.py:50)."""
    rng = np.random.default_rng(seed)
    idx = [0] * len(datasets)
    seen_all = [False] * len(datasets)
    rows = []
    p = np.asarray(probabilities, dtype=np.float64)
    p = p / p.sum()
    while True:
        k = int(rng.choice(len(datasets), p=p))
        if idx[k] >= len(datasets[k]):
            if stopping_strategy == "first_exhausted":
                break
            seen_all[k] = True
            if all(seen_all):
                break
            idx[k] = 0
        rows.append(datasets[k][idx[k]])
        idx[k] += 1
        if stopping_strategy == "first_exhausted" and any(i >= len(d) for i, d in zip(idx, datasets)):
            break
    return TokenDataset(rows)


def init_dataset(cfg, tokeniser) -> Tuple[Dict[str, object], object]:
    """hf_dataset.py:29-66: single or multi dataset, optional saved_ds_path cache (binary shards here),
    collator choice by cfg.data.packing."""
    data = _get(cfg, "data")
    saved = _get(data, "saved_ds_path", None)
    if saved and os.path.isdir(saved):
        logger.info(f"Loading dataset from {saved}")
        dataset = {s: TokenShardDataset(os.path.join(saved, s)) for s in ("train", "validation")
                   if os.path.isdir(os.path.join(saved, s))}
    else:
        tp = _get(data, "train_path")
        if isinstance(tp, (list, tuple)):
            ratios = list(_get(data, "train_ratios"))
            assert len(tp) == len(ratios), "Number of train paths should match number of train ratios"
            vp = _get(data, "val_path")
            vp = [vp] if isinstance(vp, str) else list(vp or [])
            vp = vp + [None] * (len(tp) - len(vp))
            parts = []
            for i in range(len(tp)):
                ds = parse_single_dataset(cfg, tokeniser, tp[i], vp[i])
                reps = _get(data, "repetitions", None)
                if reps:
                    ds["train"] = TokenDataset(ds["train"].rows * int(reps[i]))
                parts.append(ds)
            train = interleave_datasets([d["train"] for d in parts], ratios, seed=0,
                                        stopping_strategy=_get(data, "stopping_strategy", "first_exhausted"))
            val = TokenDataset(list(chain.from_iterable(d["validation"].rows for d in parts if "validation" in d)))
            dataset = {"train": train, "validation": val}
        else:
            dataset = parse_single_dataset(cfg, tokeniser, tp, _get(data, "val_path"))
        if saved:
            logger.info(f"Saving dataset to {saved}")
            for s, d in dataset.items():
                write_token_shard(os.path.join(saved, s), d)
    if _get(data, "packing", False):
        collator = DataCollatorWithFlattening(return_tensors="pt")
    else:
        collator = DataCollatorForLanguageModeling(tokeniser.text_tokeniser, mlm=False, return_tensors="pt")
    return dataset, collator


def init_preference_optimization_dataset(cfg) -> Dict[str, List[Dict[str, str]]]:
    """hf_dataset.py:138-148 (the auto-BLEU repetition filter needs nltk - absent - and is skipped with a warning)."""
    out = {"train": [dict(prompt=r["prompt"], chosen=r["chosen"], rejected=r["rejected"])
                     for r in _read_jsonl(glob(_get(cfg, "train_path")))]}
    if _get(cfg, "val_path", None) is not None:
        out["validation"] = [dict(prompt=r["prompt"], chosen=r["chosen"], rejected=r["rejected"])
                             for r in _read_jsonl(glob(_get(cfg, "val_path")))]
    if _get(cfg, "repetition_filter", False):
        logger.warning("repetition_filter needs nltk (not installed); rows are kept unfiltered")
    return out
