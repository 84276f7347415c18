"""The learnable token stream and step settings of the 200-step loss-curve tests (ids[t+1] = ids[t] + stride mod 500).

One definition shared by tests/golden/make_golden_traj.py (the real reference model on the HF/torch step),
tests/test_oracle_golden.py (the oracle's step restatement against that fixture) and tests/test_gpu_train.py (the engine
trainer against both), so that all three see the same rows in the same batch order."""
import torch

from slamkit_amd.data import DataCollatorForLanguageModeling, TokenDataset
from slamkit_amd.trainer.dp import seeded_batches

STEPS, BS = 200, 4
LR, WARMUP, MIN_LR, CLIP = 3e-3, 5, 5e-5, 0.5
SEED = 13   # SLAMTrainingArguments.seed of the run: the batch order


def dataset():
    g = torch.Generator().manual_seed(3)
    rows = []
    for i in range(STEPS * BS):
        n = int(torch.randint(40, 64, (1,), generator=g))
        start, stride = int(torch.randint(0, 500, (1,), generator=g)), [1, 3, 7][i % 3]
        ids = [1] + [((start + stride * t) % 500) + 2 for t in range(n)] + [1]
        rows.append({"input_ids": ids, "attention_mask": [1] * len(ids)})
    return TokenDataset(rows)


def collator():
    return DataCollatorForLanguageModeling(pad_token_id=0)


def stream():
    """The collated micro-batches of the single epoch, in the trainer's order (seeded shuffle, one rank)."""
    ds, coll = dataset(), collator()
    for b in seeded_batches(len(ds), BS, SEED, 0)[:STEPS]:
        yield coll([ds[i] for i in b])


def load_fixture():
    """tests/golden/traj.npz: the REAL reference model's curves (make_golden_traj.py)."""
    import os

    import numpy as np
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traj.npz")))


def load_oracle_curves():
    """tests/golden/traj_oracle.npz: oracle_loop's loss curves in the four settings of the GPU loss-curve test."""
    import os

    import numpy as np
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traj_oracle.npz")))


def oracle_loop(bf16_state: bool, bf16_acts: bool, round_weights: bool = False, steps: int = STEPS):
    """The oracle's restatement of the HF Trainer step (oracle/slam_oracle.py: forward_loss_grads, clip_coef,
    cosine_with_min_lr, adamw_update / adamw_update_bf16) run over the stream.
      bf16_state   : the recipe's precision - bf16 parameters, gradients rounded to bf16, bf16 AdamW state;
      bf16_acts    : the reference's bf16-autocast activation path emulated (bf16 tensors between modules);
      round_weights: compute with the bf16 rounding of fp32 master weights (what the engine does in fp32-state mode).
    Returns (losses, pre-clip gradient norms, learning rates, final parameters)."""
    from oracle import slam_oracle as O
    cfg = O.TINY
    sd = O.init_weights(cfg, seed=11, bias_std=0.0, norm_jitter=0.0)
    wdt = torch.bfloat16 if bf16_state else torch.float32
    p = {k: v.to(wdt).clone() for k, v in sd.items()}
    mo = {k: torch.zeros_like(v) for k, v in p.items()}
    vo = {k: torch.zeros_like(v) for k, v in p.items()}
    losses, gns, lrs = [], [], []
    for step, mb in enumerate(stream()):
        if step >= steps:
            break
        pw = {k: (v.to(torch.bfloat16).float() if round_weights else v.float()) for k, v in p.items()}
        l, _, gr = O.forward_loss_grads(cfg, pw, mb["input_ids"], mb["labels"], attention_mask=mb["attention_mask"],
                                        num_items_in_batch=float((mb["labels"] != -100).sum()), bf16_acts=bf16_acts)
        losses.append(float(l))
        if bf16_state:  # the reference's gradients live in the parameters' dtype
            gr = {k: v.to(torch.bfloat16).float() for k, v in gr.items()}
        tot, coef = O.clip_coef(gr, CLIP)
        gns.append(tot)
        lr = LR * O.cosine_with_min_lr(step, WARMUP, STEPS, MIN_LR / LR)
        lrs.append(lr)
        for k in p:
            if bf16_state:
                O.adamw_update_bf16(p[k], (gr[k] * coef).to(torch.bfloat16), mo[k], vo[k], step + 1, lr)
            else:
                O.adamw_update(p[k], gr[k] * coef, mo[k], vo[k], step + 1, lr)
    return losses, gns, lrs, p


def ema(x, k=0.2):
    o, a = [], x[0]
    for v in x:
        a = (1 - k) * a + k * v
        o.append(a)
    return o


def worst(x, y):
    """Largest single-step relative deviation of curve x from curve y."""
    return max(abs(float(u) - float(v)) / float(v) for u, v in zip(x, y))
