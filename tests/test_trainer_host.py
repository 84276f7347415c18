"""Host logic of the step loop (no GPU): schedule, callbacks, batch dealing."""
import time

import torch

from oracle import slam_oracle as O
from slamkit_amd.trainer.callbacks import (MaxTokensStopperCallback, RunTimeStopperCallback, TrainerControl,
                                           TrainerState, parse_run_time)
from slamkit_amd.trainer.dp import seeded_batches, shard_batches
from slamkit_amd.trainer.training_args import SLAMTrainingArguments, lr_lambda


def test_lr_schedule_matches_oracle_and_hf():
    a = SLAMTrainingArguments(warmup_steps=10, warmup_ratio=0.01)
    for s in range(0, 61):
        assert abs(lr_lambda(a, s, 60) - O.cosine_with_min_lr(s, 10, 60, 5e-5 / 1e-3)) < 1e-12
    from transformers.optimization import get_scheduler
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    sch = get_scheduler("cosine_with_min_lr", opt, num_warmup_steps=10, num_training_steps=60,
                        scheduler_specific_kwargs={"min_lr": 5e-5})
    for s in range(60):
        assert abs(sch.get_last_lr()[0] - 1e-3 * lr_lambda(a, s, 60)) < 1e-12
        opt.step()
        sch.step()
    # warmup_ratio takes over when warmup_steps == 0 (cli/train.py:48-54 zeroes it in that case)
    b = SLAMTrainingArguments(warmup_steps=0, warmup_ratio=0.1)
    assert b.get_warmup_steps(200) == 20 and abs(lr_lambda(b, 10, 200) - 0.5) < 1e-12


def test_run_time_parsing_and_stoppers():
    assert parse_run_time("1-02:03:04") == 86400 + 7384 and parse_run_time("00:00:05") == 5 and parse_run_time(9) == 9
    st, ctl = TrainerState(), TrainerControl()
    cb = MaxTokensStopperCallback(1000)
    cb.on_train_begin(None, st, ctl)
    st.num_input_tokens_seen = 999
    cb.on_step_end(None, st, ctl)
    assert not ctl.should_training_stop
    st.num_input_tokens_seen = 1000
    cb.on_step_end(None, st, ctl)
    assert ctl.should_training_stop and ctl.should_save and ctl.should_evaluate
    ctl2 = TrainerControl()
    rt = RunTimeStopperCallback(0)
    rt.on_train_begin(None, st, ctl2)
    time.sleep(0.01)
    rt.on_step_end(None, st, ctl2)
    assert ctl2.should_training_stop


def test_batch_dealing_round_robin_like_accelerate():
    batches = seeded_batches(37, 4, seed=42, epoch=0)
    assert sorted(i for b in batches for i in b) == list(range(37)) and len(batches) == 10
    assert seeded_batches(37, 4, 42, 0) == batches and seeded_batches(37, 4, 42, 1) != batches
    for world in (2, 3, 4, 8):
        per = [shard_batches(batches, r, world) for r in range(world)]
        assert len({len(p) for p in per}) == 1            # even_batches: same step count on every rank
        full = (len(batches) // world) * world
        for r in range(world):
            assert per[r][: full // world] == batches[r:full:world]   # rank r takes r, r+world, ...
        seen = [tuple(b) for p in per for b in p]
        assert set(map(tuple, batches)) <= set(seen)       # nothing dropped
