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


# ---- binary shard format: tokens.bin (little-endian uint16, or uint32 when an id does not fit - the interleaved
#      speech-text vocabulary reaches 152,166; all sequences back to back) + index.npy (int64 offsets, n+1 entries) +
#      meta.json (records the dtype). ids = unit + 2 exactly as UnitTokeniser produces them. ---------------------------
def write_token_shard(path: str, dataset) -> None:
    n = len(dataset)
    lens = np.fromiter((len(dataset[i]["input_ids"]) for i in range(n)), dtype=np.int64, count=n)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    buf = np.empty(int(off[-1]), dtype=np.int64)
    for i in range(n):
        buf[off[i]:off[i + 1]] = dataset[i]["input_ids"]
    if buf.size and int(buf.min()) < 0:
        raise ValueError("negative token id")
    top = int(buf.max()) if buf.size else 0
    if top > 0xFFFFFFFF:
        raise ValueError("token id does not fit uint32")
    dt = "uint16" if top <= 0xFFFF else "uint32"
    # written next to the target and renamed into place: a reader (or a concurrent rank) never sees half a shard
    tmp = f"{path}.tmp{os.getpid()}"
    os.makedirs(tmp, exist_ok=True)
    buf.astype("<u2" if dt == "uint16" else "<u4").tofile(os.path.join(tmp, "tokens.bin"))
    np.save(os.path.join(tmp, "index.npy"), off)
    with open(os.path.join(tmp, "meta.json"), "w") as f:
        json.dump({"format": "slam-token-shard-v1", "dtype": dt, "sequences": int(n), "tokens": int(off[-1])}, f)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if os.path.isdir(path):
        import shutil
        shutil.rmtree(path)
    os.rename(tmp, path)


class TokenShardDataset:
    """Memory-mapped view of a shard written by write_token_shard."""

    def __init__(self, path: str):
        self.off = np.load(os.path.join(path, "index.npy"))
        dt = "uint16"
        meta = os.path.join(path, "meta.json")
        if os.path.exists(meta):
            with open(meta) as f:
                dt = json.load(f).get("dtype", "uint16")
        if dt not in ("uint16", "uint32"):
            raise ValueError(f"{path}: unknown shard dtype {dt!r}")
        self.tok = np.memmap(os.path.join(path, "tokens.bin"), dtype="<u2" if dt == "uint16" else "<u4", mode="r")
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


def interleave_indices(lengths: Sequence[int], probabilities: Optional[Sequence[float]], seed: Optional[int] = 0,
                       stopping_strategy: str = "first_exhausted") -> List[Tuple[int, int]]:
    """(dataset, row) pairs in the order `datasets.interleave_datasets` selects them for map-style datasets
    (site-packages datasets/arrow_dataset.py `_interleave_map_style_datasets`; called at hf_dataset.py:44-55 with
    seed=0): source ids are drawn 1000 at a time from np.random.default_rng(seed).choice(n, size=1000, p=probabilities);
    a source that runs out is flagged exhausted and restarts from its first row; the walk stops BEFORE the draw that
    follows the stopping condition (any source exhausted / all sources exhausted). Pinned against the library by
    tests/golden/interleave_ds.json."""
    n = len(lengths)
    if stopping_strategy not in ("first_exhausted", "all_exhausted", "all_exhausted_without_replacement"):
        raise ValueError(f"{stopping_strategy} is not supported")
    lengths = [int(x) for x in lengths]
    if probabilities is None:
        if stopping_strategy == "first_exhausted":
            return [(k, i) for i in range(min(lengths)) for k in range(n)]
        if stopping_strategy == "all_exhausted":
            return [(k, i % lengths[k]) for i in range(max(lengths)) for k in range(n)]
        return [(k, i) for i in range(max(lengths)) for k in range(n) if i < lengths[k]]
    exhausted = np.full(n, False)
    done = np.any if stopping_strategy == "first_exhausted" else np.all
    without_replacement = stopping_strategy == "all_exhausted_without_replacement"
    rng = np.random.default_rng(seed)
    cur = [0] * n
    out: List[Tuple[int, int]] = []
    while True:
        for k in rng.choice(n, size=1000, p=probabilities):
            k = int(k)
            if done(exhausted):
                return out
            if not without_replacement or not exhausted[k]:
                out.append((k, cur[k]))
                cur[k] += 1
            if cur[k] >= lengths[k]:
                exhausted[k] = True
                if not without_replacement:
                    cur[k] = 0


def interleave_datasets(datasets: List[TokenDataset], probabilities: Optional[List[float]] = None, seed: Optional[int] = 0,
                        stopping_strategy: str = "first_exhausted") -> TokenDataset:
    """datasets.interleave_datasets on in-memory TokenDatasets: same rows in the same order (interleave_indices)."""
    order = interleave_indices([len(d) for d in datasets], probabilities, seed, stopping_strategy)
    return TokenDataset([datasets[k][i] for k, i in order])


def init_dataset(cfg, tokeniser) -> Tuple[Dict[str, object], object]:
    """hf_dataset.py:29-66: single or multi dataset, optional saved_ds_path cache (binary shards here),
    collator choice by cfg.data.packing."""
    data = _get(cfg, "data")
    saved = _get(data, "saved_ds_path", None)
    # load-or-build is decided ONCE (rank 0, on the completion marker written after the last shard) and broadcast: ranks
    # that looked at the directory themselves could see a half-written cache, take different branches and miss the barrier
    have_cache = bool(saved) and os.path.isfile(os.path.join(saved, "_COMPLETE"))
    # a directory with shards but no marker was written before the marker existed, or by an interrupted run: the reference
    # treats an existing directory as the cache (hf_dataset.py:31-33) - re-tokenising over it would mix old and new shards
    # (every split directory counts: a cache holding only validation/ must not be rebuilt over either)
    def _has_shards(split):
        d = os.path.join(saved, split)
        return os.path.isdir(d) and bool(os.listdir(d))
    legacy = bool(saved) and not have_cache and any(_has_shards(sp) for sp in ("train", "validation"))
    # opt-in (data.accept_unmarked_cache or SLAM_ACCEPT_UNMARKED_CACHE=1): load a marker-less directory as the cache, with a
    # warning - what the reference does with any existing directory, and what caches written before the marker existed need
    accept = bool(_get(data, "accept_unmarked_cache", False)) or os.environ.get("SLAM_ACCEPT_UNMARKED_CACHE", "0") == "1"
    if legacy and accept and _has_shards("train"):
        logger.warning(f"{saved} has token shards but no _COMPLETE marker: loading it as the cache because accept_unmarked_cache is set")
        have_cache, legacy = True, False
    import torch.distributed as _dist
    _multi = _dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1
    if _multi:
        flag = [have_cache, legacy]
        _dist.broadcast_object_list(flag, src=0)
        have_cache, legacy = bool(flag[0]), bool(flag[1])
    if legacy:
        raise RuntimeError(f"{saved} holds token shards but no _COMPLETE marker: either an earlier version wrote it (verify it, then "
                           f"`touch {os.path.join(saved, '_COMPLETE')}` or set data.accept_unmarked_cache / SLAM_ACCEPT_UNMARKED_CACHE=1 "
                           f"to accept it as the cache) or a run was interrupted while writing it (delete the directory to rebuild)")
    if saved and have_cache:
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
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            if not multi or dist.get_rank() == 0:  # one writer; the other ranks wait, then every rank maps the shards
                logger.info(f"Saving dataset to {saved}")
                for s, d in dataset.items():
                    write_token_shard(os.path.join(saved, s), d)
                with open(os.path.join(saved, "_COMPLETE"), "w") as f:  # written last: the cache is valid only with it
                    f.write("ok\n")
            if multi:
                dist.barrier()
    if _get(data, "packing", False):
        collator = DataCollatorWithFlattening(return_tensors="pt")
    else:
        collator = DataCollatorForLanguageModeling(tokeniser.text_tokeniser, mlm=False, return_tensors="pt")
    return dataset, collator


_TB_RULES = None


def _treebank_rules():
    """The substitution passes of nltk's NLTKWordTokenizer (nltk/tokenize/destructive.py; the reference constructs it in
    slamkit/data/hf_dataset.py:128-129 and does not pin an nltk version). nltk is not installed in this image, so the
    published rule set is restated here: every pass pads what it detaches with spaces, the text is split on whitespace at
    the end. Order matters and is nltk's: opening quotes, punctuation, brackets, double dashes, closing quotes, clitics."""
    import re
    c = re.compile
    opening = [
        (c("([\u00ab\u201c\u2018\u201e]|[`]+)"), r" \1 "),
        (c(r'^"'), "``"),
        (c(r"(``)"), r" \1 "),
        (c(r"([ \(\[{<])(\"|'{2})"), r"\1 `` "),
        (c(r"(?i)(')(?!re|ve|ll|m|t|s|d|n)(\w)\b"), r"\1 \2"),
    ]
    punct = [
        (c("([^\\.])(\\.)([\\]\\)}>\"'\u00bb\u201d\u2019 ]*)\\s*$"), r"\1 \2 \3 "),  # only the text-final period is detached
        (c(r"([:,])([^\d])"), r" \1 \2"),
        (c(r"([:,])$"), r" \1 "),
        (c(r"\.{2,}"), r" \g<0> "),
        (c(r"[;@#$%&]"), r" \g<0> "),
        (c(r"([^\.])(\.)([\]\)}>\"']*)\s*$"), r"\1 \2\3 "),
        (c(r"[?!]"), r" \g<0> "),
        (c(r"([^'])' "), r"\1 ' "),
        (c(r"[*]"), r" \g<0> "),
    ]
    brackets = (c(r"[\]\[\(\)\{\}\<\>]"), r" \g<0> ")
    dashes = (c(r"--"), r" -- ")
    closing = [
        (c("([\u00bb\u201d\u2019])"), r" \1 "),
        (c(r"''"), " '' "),
        (c(r'"'), " '' "),
        (c(r"([^' ])('[sS]|'[mM]|'[dD]|') "), r"\1 \2 "),
        (c(r"([^' ])('ll|'LL|'re|'RE|'ve|'VE|n't|N'T) "), r"\1 \2 "),
    ]
    clitics = [c(p) for p in (r"(?i)\b(can)(not)\b", r"(?i)\b(d)('ye)\b", r"(?i)\b(gim)(me)\b", r"(?i)\b(gon)(na)\b",
                              r"(?i)\b(got)(ta)\b", r"(?i)\b(lem)(me)\b", r"(?i)\b(more)('n)\b", r"(?i)\b(wan)(na)(?=\s)",
                              r"(?i) ('t)(is)\b", r"(?i) ('t)(was)\b")]
    return opening, punct, brackets, dashes, closing, clitics


def word_tokenize(text: str) -> List[str]:
    """Word splitter of the auto-BLEU repetition filter: NLTKWordTokenizer().tokenize(text) (calculation_utils.py:32-35),
    restated from nltk's published rules (see _treebank_rules). Pinned on the known answers of nltk's own documentation
    (tests/test_data_pipeline.py); no nltk in this image to generate further vectors - said so in DESIGN.md."""
    global _TB_RULES
    if _TB_RULES is None:
        _TB_RULES = _treebank_rules()
    opening, punct, brackets, dashes, closing, clitics = _TB_RULES
    for rx, sub in opening:
        text = rx.sub(sub, text)
    for rx, sub in punct:
        text = rx.sub(sub, text)
    text = brackets[0].sub(brackets[1], text)
    text = dashes[0].sub(dashes[1], text)
    text = " " + text + " "
    for rx, sub in closing:
        text = rx.sub(sub, text)
    for rx in clitics:
        text = rx.sub(r" \1 \2 ", text)
    return text.split()


def calc_ngram(text: str, n: int) -> List[str]:
    tokens = word_tokenize(text)
    return [" ".join(tokens[i:i + n]) for i in range(len(tokens) - n + 1)]


def calc_auto_bleu(text: str, n: int) -> float:
    """Fraction of n-grams that occur more than once in the text (slamkit/utils/calculation_utils.py:37-47)."""
    ngrams = calc_ngram(text, n)
    if not ngrams:
        return 0
    from collections import Counter
    cnt = Counter(ngrams)
    return sum(1 for g in ngrams if cnt[g] > 1) / len(ngrams)


def get_repetition_filter_fn(auto_bleu_n: int, max_auto_bleu: float):
    """hf_dataset.py:125-133: keep a pair when auto-BLEU(prompt_text + " " + chosen_text) < max_auto_bleu."""
    return lambda x: calc_auto_bleu(x["prompt_text"] + " " + x["chosen_text"], auto_bleu_n) < max_auto_bleu


def init_preference_optimization_dataset(cfg) -> Dict[str, List[Dict[str, str]]]:
    """hf_dataset.py:138-148: jsonl rows -> optional repetition filter (on the raw rows, which carry prompt_text /
    chosen_text) -> only the prompt / chosen / rejected columns."""
    keep = (lambda x: True)
    if _get(cfg, "repetition_filter", False):
        keep = get_repetition_filter_fn(int(_get(cfg, "auto_bleu_n", 2)), float(_get(cfg, "max_auto_bleu", 0.3)))
    cols = lambda rows: [dict(prompt=r["prompt"], chosen=r["chosen"], rejected=r["rejected"]) for r in rows if keep(r)]  # noqa: E731
    out = {"train": cols(_read_jsonl(glob(_get(cfg, "train_path"))))}
    if _get(cfg, "val_path", None) is not None:
        out["validation"] = cols(_read_jsonl(glob(_get(cfg, "val_path"))))
    return out
