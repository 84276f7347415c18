"""Golden index streams of HuggingFace `datasets.interleave_datasets` (the call at /root/reference
slamkit/data/hf_dataset.py:44-55: probabilities=train_ratios, seed=0, stopping_strategy from the config), generated
with the installed `datasets` library in the authoring container:
    python tests/golden/make_golden_interleave_ds.py   -> tests/golden/interleave_ds.json
Each case: dataset lengths, probabilities, seed, strategy -> the (dataset, row) order of the interleaved result."""
import json
import os

import datasets
from datasets import Dataset, interleave_datasets

HERE = os.path.dirname(os.path.abspath(__file__))
cases = []
for lengths, probs, seed, strat in [
    ([7, 11], [0.5, 0.5], 0, "first_exhausted"),
    ([40, 25, 60], [0.6, 0.3, 0.1], 0, "first_exhausted"),
    ([40, 25, 60], [0.6, 0.3, 0.1], 0, "all_exhausted"),
    ([1500, 900], [0.7, 0.3], 0, "first_exhausted"),      # crosses the 1000-draw batches of the library
    ([1500, 900], [0.7, 0.3], 0, "all_exhausted"),
    ([30, 50], [0.25, 0.75], 3, "all_exhausted"),
    ([5, 9, 4], None, 0, "first_exhausted"),
    ([5, 9, 4], None, 0, "all_exhausted"),
    ([13, 21], [0.5, 0.5], 0, "all_exhausted_without_replacement"),
]:
    dss = [Dataset.from_dict({"k": [k] * n, "i": list(range(n))}) for k, n in enumerate(lengths)]
    out = interleave_datasets(dss, probabilities=probs, seed=seed, stopping_strategy=strat)
    cases.append({"lengths": lengths, "probabilities": probs, "seed": seed, "stopping_strategy": strat,
                  "order": [[int(a), int(b)] for a, b in zip(out["k"], out["i"])]})
with open(os.path.join(HERE, "interleave_ds.json"), "w") as f:
    json.dump({"datasets_version": datasets.__version__, "cases": cases}, f)
print([(c["lengths"], c["stopping_strategy"], len(c["order"])) for c in cases])
