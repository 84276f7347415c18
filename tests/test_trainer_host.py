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


def test_collate_prefetch_thread_yields_the_synchronous_stream():
    """dataloader_num_workers > 0: the background collate thread hands over exactly the micro-batch groups of the
    synchronous path, in order, also when the consumer stops early, and surfaces a collate error in the consumer."""
    import types
    from slamkit_amd.trainer.slam_trainer import SLAMTrainer

    def make(workers, fail_at=None):
        calls = []

        def collate(idxs):
            if fail_at is not None and len(calls) == fail_at:
                raise ValueError("bad row")
            calls.append(list(idxs))
            return {"input_ids": torch.tensor(idxs)[None], "labels": torch.tensor(idxs)[None]}
        return types.SimpleNamespace(args=types.SimpleNamespace(dataloader_num_workers=workers), _collate=collate), calls

    batches = [[4 * i + j for j in range(4)] for i in range(11)]
    ref = [[m["input_ids"].tolist() for m in g] for g in SLAMTrainer._micro_batches(make(0)[0], batches, 3)]
    assert [len(g) for g in ref] == [3, 3, 3, 2]
    got = [[m["input_ids"].tolist() for m in g] for g in SLAMTrainer._micro_batches(make(2)[0], batches, 3)]
    assert got == ref
    fake, calls = make(2)
    it = SLAMTrainer._micro_batches(fake, batches, 3)
    first = next(it)
    it.close()  # early stop: the producer thread is released, not left blocked on a full queue
    assert [m["input_ids"].tolist() for m in first] == ref[0] and len(calls) <= 9
    import pytest
    with pytest.raises(ValueError, match="bad row"):
        list(SLAMTrainer._micro_batches(make(2, fail_at=4)[0], batches, 3))
