"""Golden vectors for the interleaved speech-text tokeniser (BASELINE configs[3] data side), produced by
importing the REAL reference (/root/reference slamkit/tokeniser/interleaving_tokeniser.py) in the authoring
container over a tiny LOCAL text tokenizer (the hub is unreachable, the reference default is facebook/opt-125m):
    HF_HUB_OFFLINE=1 python tests/golden/make_golden_interleave.py
Writes tests/golden/tiny_text_tokenizer/ (the tokenizer files, data) and tests/golden/interleave.json.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
stub = tempfile.mkdtemp()
os.makedirs(os.path.join(stub, "omegaconf"))
with open(os.path.join(stub, "omegaconf", "__init__.py"), "w") as f:
    f.write("class DictConfig(dict): pass\nclass ListConfig(list): pass\nclass OmegaConf: pass\n")
sys.path[:0] = [stub, REF]

from tokenizers import Tokenizer, models, pre_tokenizers, processors  # noqa: E402
from transformers import PreTrainedTokenizerFast  # noqa: E402
from slamkit.tokeniser.interleaving_tokeniser import InterleavingTokeniser  # noqa: E402

# ---- tiny word-level text tokenizer with a BOS-prepending post-processor (the shape of OPT's behaviour) -----
words = ("the quick brown fox jumps over lazy dog a speech language model reads and listens to every word it "
         "hears then answers in text or in sound hello world").split()
vocab = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3}
for w in words:
    vocab.setdefault(w, len(vocab))
tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
tk.pre_tokenizer = pre_tokenizers.Whitespace()
tk.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
fast = PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>")
tdir = os.path.join(HERE, "tiny_text_tokenizer")
os.makedirs(tdir, exist_ok=True)
fast.save_pretrained(tdir)


class StubFE:  # only what _create_interleaved_text needs from the feature extractor (mhubert_25: 25 units / s)
    def get_unit_duration(self):
        return 0.04


def make_rep(seed, n_words):
    g = np.random.RandomState(seed)
    units, dur = [], []
    aligned, t = [], 0.0
    for i in range(n_words):
        w = words[int(g.randint(len(words)))]
        wl = float(g.uniform(0.12, 0.45))
        aligned.append([(" " if i else "") + w, round(t, 3), round(t + wl, 3)])
        t += wl + float(g.uniform(0.0, 0.08))
    total = int(t / 0.04) + 3
    while sum(dur) < total:
        units.append(int(g.randint(500)))
        dur.append(int(g.randint(1, 4)))
    return {"units": units, "duration": dur, "aligned_text": aligned}


reps = [make_rep(1, 9), make_rep(2, 23), make_rep(3, 1), make_rep(4, 40)]
out = {"reps": reps, "cases": []}
for method, span, prob in (("random", None, None), ("span", 3, 0.3), ("poisson", 4, 0.3), ("poisson", 10, 0.3)):
    tok = InterleavingTokeniser(StubFE(), num_units=500, load_fe=True, text_tokeniser_path=tdir,
                                interleave_method=method, interleave_span=span, interleave_prob=prob)
    for seed in (0, 7):
        torch.manual_seed(seed)
        np.random.seed(seed)
        strings = tok.stringify_representation(reps, mode="train")
        ids = [list(tok.prepare_sample({"audio_repr": s})["input_ids"]) for s in strings]
        out["cases"].append({"method": method, "span": span, "prob": prob, "seed": seed, "strings": strings, "input_ids": ids})
tok = InterleavingTokeniser(StubFE(), num_units=500, load_fe=True, text_tokeniser_path=tdir)
out["test_mode_strings"] = tok.stringify_representation(reps[:2], mode="test")
out["vocab_size"] = len(tok.text_tokeniser)
out["ignore_speech"] = tok.get_ignore_tokens("SPEECH")
out["ignore_text"] = tok.get_ignore_tokens("TEXT")
out["speech_text_ids"] = [tok.text_tokeniser.encode("<speech>")[0], tok.text_tokeniser.encode("<text>")[0]]
sample = torch.tensor(out["cases"][4]["input_ids"][1])
out["decode_speech"] = tok.decode_sample(sample, "SPEECH").tolist()
out["decode_text"] = tok.decode_sample(sample, "TEXT")
with open(os.path.join(HERE, "interleave.json"), "w") as f:
    json.dump(out, f)
print("vocab", out["vocab_size"], "cases", len(out["cases"]))
print(out["cases"][4]["strings"][0][:300])
print(out["cases"][4]["input_ids"][0][:40])
print("decode_text:", out["decode_text"][:120])
