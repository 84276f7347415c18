"""World-size-2 (gloo, CPU) run of the REAL SLAMTrainer step loop - token-count all-reduce, both
`average_tokens_across_devices` modes, round-robin batch dealing, bucket callback -> GradBucketReducer, joint
stop / evaluate / save decisions - around a stub model that implements the engine surface the trainer calls
(forward / backward with bucket callback / grad_norm / adamw_step) with a bag-of-embeddings LM in plain torch
(the HIP engine needs a GPU; the oracle's clip and AdamW are the stub's optimizer: test infrastructure).
Checked: after 3 optimizer steps (GA 2) every rank holds the parameters of a single-process run over the same
global batches, to fp32 round-off; the logged loss is the global token-mean loss."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

V, H = 37, 16


class StubEngine:
    def __init__(self, m):
        self.m = m
        self.n_params = m.flat_master.numel()

    def set_option(self, key, value):
        if key == "grad_overwrite_next":
            self.m.overwrite = bool(value)

    def join(self):
        pass

    def grad_norm(self, max_norm, norm_out):
        from oracle import slam_oracle as O
        n, c = O.clip_coef({"g": self.m.flat_grads}, max_norm)
        norm_out[0], norm_out[1] = float(n), float(c)

    def adamw_step(self, master, m, v, norm_out, lr, b1, b2, eps, wd, step, zero_grad=True):
        from oracle import slam_oracle as O
        O.adamw_update(master, self.m.flat_grads * float(norm_out[1]), m, v, step, lr, b1, b2, eps, wd)
        if zero_grad:
            self.m.flat_grads.zero_()


class StubLM:
    """logits[b,t] = E[ids[b,t]] @ W^T ; loss = compute_loss (shifted CE, sum / num_items)."""

    def __init__(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.device = torch.device("cpu")
        self.flat_master = torch.randn(V * H + V * H, generator=g) * 0.3
        self.flat_grads = torch.zeros_like(self.flat_master)
        self.engine = StubEngine(self)
        self.overwrite = False
        self.saved = []

    def forward(self, input_ids=None, labels=None, num_items_in_batch=None, **kw):
        p = self.flat_master.detach().clone().requires_grad_(True)
        E, W = p[: V * H].view(V, H), p[V * H:].view(V, H)
        logits = E[input_ids] @ W.t()
        from oracle.slam_oracle import compute_loss
        loss = compute_loss(logits, labels, num_items_in_batch=num_items_in_batch)
        self._graph = (loss, p)
        return type("Out", (), {"loss": loss.detach(), "logits": logits.detach()})()

    def backward(self, grad_scale=1.0, bucket_layers=0, bucket_cb=None):
        loss, p = self._graph
        (g,) = torch.autograd.grad(loss * grad_scale, p)
        if self.overwrite:
            self.flat_grads.copy_(g)
            self.overwrite = False
        else:
            self.flat_grads.add_(g)
        if bucket_cb is not None:  # ranges reported back to front, like slam_backward
            bucket_cb(V * H, V * H)
            bucket_cb(0, V * H)

    def zero_grad(self):
        self.flat_grads.zero_()

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        self.saved.append(path)


def make_rows(n=24, seed=3):
    g = torch.Generator().manual_seed(seed)
    rows = []
    for _ in range(n):
        k = int(torch.randint(3, 12, (1,), generator=g))
        rows.append({"input_ids": torch.randint(1, V, (k,), generator=g).tolist(), "attention_mask": [1] * k})
    return rows


def run_trainer(rank, world, avg_tokens, out_dir):
    from slamkit_amd.data import DataCollatorForLanguageModeling
    from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments
    args = SLAMTrainingArguments(output_dir=out_dir, per_device_train_batch_size=2, gradient_accumulation_steps=2,
                                 learning_rate=1e-2, warmup_steps=1, warmup_ratio=0.0, max_steps=3, logging_steps=1,
                                 average_tokens_across_devices=avg_tokens, ddp_bucket_layers=1, seed=5, save_steps=0, ddp_comm_dtype="float32")
    model = StubLM()
    tr = SLAMTrainer(model=model, args=args, data_collator=DataCollatorForLanguageModeling(pad_token_id=0),
                     train_dataset=make_rows())
    assert (tr.rank, tr.world) == (rank, world)
    tr.train()
    return model, tr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        for avg in (True, False):
            model, tr = run_trainer(rank, world, avg, os.path.join(tmp, f"r{rank}"))
            res[avg] = (model.flat_master.tolist(), [h["loss"] for h in tr.state.log_history],
                        tr.state.num_input_tokens_seen)
        # joint control decisions: only rank 1's clock fired -> both ranks must evaluate / save / stop
        c = tr.control
        c.should_training_stop = c.should_evaluate = c.should_save = (rank == 1)
        tr._sync_control()
        flags = (c.should_training_stop, c.should_evaluate, c.should_save)
        c.should_training_stop = c.should_evaluate = c.should_save = False
        tr._sync_control()
        flags0 = (c.should_training_stop, c.should_evaluate, c.should_save)
        q.put((rank, res, flags, flags0))
    finally:
        dist.destroy_process_group()


def _single_process_reference(avg_tokens, world=2):
    """The same 3 optimizer steps in one process: at each step the union of what the ranks see."""
    from oracle import slam_oracle as O
    from slamkit_amd.data import DataCollatorForLanguageModeling
    from slamkit_amd.trainer.dp import seeded_batches, shard_batches
    from slamkit_amd.trainer.training_args import SLAMTrainingArguments, lr_lambda
    rows, coll = make_rows(), DataCollatorForLanguageModeling(pad_token_id=0)
    m = StubLM()
    a = SLAMTrainingArguments(learning_rate=1e-2, warmup_steps=1, warmup_ratio=0.0)
    ea, eq = torch.zeros_like(m.flat_master), torch.zeros_like(m.flat_master)
    per_rank = [shard_batches(seeded_batches(len(rows), 2, 5, 0), r, world) for r in range(world)]
    losses, seen = [], 0
    for step in range(3):
        micro = [[coll([rows[i] for i in per_rank[r][2 * step + j]]) for j in range(2)] for r in range(world)]
        counts = [sum(int((mb["labels"] != -100).sum()) for mb in micro[r]) for r in range(world)]
        m.flat_grads.zero_()
        tot = 0.0
        for r in range(world):
            for mb in micro[r]:
                n = sum(counts) if avg_tokens else counts[r]
                out = m.forward(input_ids=mb["input_ids"], labels=mb["labels"], num_items_in_batch=float(n))
                m.backward(1.0 if avg_tokens else 1.0 / world)
                tot += float(out.loss) * (1.0 if avg_tokens else 1.0 / world)
        losses.append(tot)
        seen += sum(counts)
        n, c = O.clip_coef({"g": m.flat_grads}, a.max_grad_norm)
        O.adamw_update(m.flat_master, m.flat_grads * float(c), ea, eq, step + 1, 1e-2 * lr_lambda(a, step, 3))
    return m.flat_master, losses, seen


def test_slam_trainer_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for avg in (True, False):
        ref_p, ref_losses, ref_seen = _single_process_reference(avg)
        for rank, r, flags, flags0 in res:
            p, losses, seen = r[avg]
            p = torch.tensor(p)
            assert torch.allclose(p, ref_p, rtol=1e-5, atol=1e-6), (avg, rank, float((p - ref_p).abs().max()))
            assert seen == ref_seen
            assert len(losses) == 3 and all(abs(a - b) < 1e-5 for a, b in zip(losses, ref_losses)), (avg, losses, ref_losses)
    for rank, _, flags, flags0 in res:
        assert flags == (True, True, True) and flags0 == (False, False, False)
    assert res[0][1][True][0] == res[1][1][True][0]  # both ranks hold the same parameters, bit for bit


# ---- early exit with a collective still posted (round-5 advisor finding) -----------------------------------------------------------
def _early_worker(rank, world, port, q, tmp):
    """max_steps = 2 of the 3 optimizer steps an epoch holds: when the loop leaves, the token-count all-reduce of step 3 has
    been POSTED (one step ahead) and not consumed. Evaluate + save fire at that same step and a stopper callback fires on rank 1
    only: the drain must come before their collectives (evaluate's all-reduce, save's barrier, the control sync) or the ranks
    pair mismatching collectives and hang / corrupt the counts. (Over gloo an undrained work object completes by itself, so
    this checks the early-exit path end to end - no hang, same step, same state, every later collective paired - rather than
    failing on the missing wait alone.)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from slamkit_amd.data import DataCollatorForLanguageModeling
        from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments
        from slamkit_amd.trainer.callbacks import TrainerCallback

        class StopOnRank1(TrainerCallback):
            def on_step_end(self, args, state, control, **kw):
                if rank == 1 and state.global_step == 2:
                    control.should_training_stop = True

        out = {}
        for name, kw in (("max_steps", dict(max_steps=2, save_steps=2, eval_strategy="steps", eval_steps=2)),
                         ("stopper", dict(max_steps=-1, num_train_epochs=1.0, save_steps=2, eval_strategy="steps", eval_steps=2))):
            args = SLAMTrainingArguments(output_dir=os.path.join(tmp, f"{name}_r{rank}"), per_device_train_batch_size=2,
                                         gradient_accumulation_steps=2, learning_rate=1e-2, warmup_steps=1, warmup_ratio=0.0,
                                         logging_steps=1, ddp_bucket_layers=1, seed=5, ddp_comm_dtype="float32", **kw)
            model = StubLM()
            type(model)._weights = property(lambda self: self.flat_master)
            rows = make_rows()
            tr = SLAMTrainer(model=model, args=args, data_collator=DataCollatorForLanguageModeling(pad_token_id=0), train_dataset=rows,
                             eval_dataset=rows[:6], callbacks=[StopOnRank1()] if name == "stopper" else [])
            posted = []
            orig = tr.post_counts
            tr.post_counts = lambda a, b: (posted.append(1), orig(a, b))[1]
            tr.train()
            ev = [h for h in tr.state.log_history if "eval_loss" in h]
            out[name] = (tr.state.global_step, model.flat_master.tolist(), len(posted), [e["eval_loss"] for e in ev], len(model.saved),
                         tr.state.num_input_tokens_seen)
        # the group is still usable and in step: one more collective pairs up
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        q.put((rank, out, float(t)))
    finally:
        dist.destroy_process_group()


def test_early_exit_drains_the_posted_count_collective_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_early_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))   # a hang here IS the regression
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, o0, s0), (r1, o1, s1) = res
    assert s0 == s1 == 3.0
    for name in ("max_steps", "stopper"):
        step0, p0, posted0, ev0, saved0, seen0 = o0[name]
        step1, p1, posted1, ev1, saved1, seen1 = o1[name]
        assert step0 == step1 == 2, (name, step0, step1)            # both ranks stop at the same step (rank 1's stopper included)
        assert p0 == p1                                               # identical parameters
        assert posted0 == posted1 == 3                                # the collective of the step that never ran WAS posted ...
        assert ev0 == ev1 and len(ev0) == 1                          # ... and evaluate's all-reduce still paired up (same global loss)
        assert saved0 == 1 and saved1 == 0                           # rank 0 saved, both passed the barrier
        assert seen0 == seen1 > 0
