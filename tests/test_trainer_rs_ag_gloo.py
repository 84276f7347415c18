"""ddp_algo = "rs_ag" (reduce-scatter gradients -> AdamW on the owned 1/N shard -> all-gather parameters) against
ddp_algo = "all_reduce" on two gloo ranks, through the REAL SLAMTrainer / ShardedGradReducer code around the stub model of
test_trainer_dp_gloo.py (the HIP engine needs a GPU). The stub implements the engine's chunked gradient-norm contract
(include/slam_engine.h: slam_grad_sumsq_chunks / slam_grad_norm_from_chunks) in torch, so the check is the one the engine
is built for: after 3 optimizer steps (GA 2) BOTH algorithms leave bit-identical parameters on both ranks, fp32 and bf16
gradient exchange alike, and the sharded optimizer state, gathered for a checkpoint, equals the replicated one."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import test_trainer_dp_gloo as base

CHUNK = 16  # elements per gradient-norm chunk of the stub (the engine: 8192)


class ChunkedStubEngine(base.StubEngine):
    def grad_chunk_info(self):
        return CHUNK, (self.n_params + CHUNK - 1) // CHUNK

    def _sums(self, off, cnt, out):
        g = self.m.flat_grads
        assert off % CHUNK == 0 and ((off + cnt) % CHUNK == 0 or off + cnt == self.n_params)
        for k in range(off // CHUNK, (off + cnt + CHUNK - 1) // CHUNK):
            out[k] = (g[k * CHUNK: min((k + 1) * CHUNK, self.n_params)] ** 2).sum()

    def grad_sumsq_chunks(self, off, cnt, out):
        self._sums(off, cnt, out)

    def grad_norm_from_chunks(self, cs, max_norm, norm_out):
        nrm = float(cs.double().sum().sqrt().float())
        norm_out[0] = nrm
        norm_out[1] = min(1.0, max_norm / (nrm + 1e-6)) if max_norm > 0 else 1.0

    def grad_norm(self, max_norm, norm_out):  # the replicated step: the same chunk sums over the whole buffer
        cs = torch.zeros(self.grad_chunk_info()[1])
        self._sums(0, self.n_params, cs)
        self.grad_norm_from_chunks(cs, max_norm, norm_out)

    def adamw_range(self, off, cnt, master, m, v, norm_out, lr, b1, b2, eps, wd, step, zero_grad=False):
        from oracle import slam_oracle as O
        sl = slice(off, off + cnt)
        O.adamw_update(master[sl], self.m.flat_grads[sl] * float(norm_out[1]), m[sl], v[sl], step, lr, b1, b2, eps, wd)
        self.m.flat_params[sl] = master[sl]

    def adamw_step(self, master, m, v, norm_out, lr, b1, b2, eps, wd, step, zero_grad=True):
        self.adamw_range(0, self.n_params, master, m, v, norm_out, lr, b1, b2, eps, wd, step)
        if zero_grad:
            self.m.flat_grads.zero_()

    def zero_grads(self):
        self.m.flat_grads.zero_()

    def add_param_wait(self, off, cnt, ev):
        pass


class ShardStubLM(base.StubLM):
    """forward reads flat_params (what the engine's kernels read: the bf16 copy the optimizer writes), not the master."""

    def __init__(self, seed=0):
        super().__init__(seed)
        self.flat_params = self.flat_master.clone()
        self.engine = ChunkedStubEngine(self)

    def forward(self, input_ids=None, labels=None, num_items_in_batch=None, **kw):
        keep = self.flat_master
        self.flat_master = self.flat_params
        try:
            return super().forward(input_ids=input_ids, labels=labels, num_items_in_batch=num_items_in_batch, **kw)
        finally:
            self.flat_master = keep


def run(rank, world, algo, comm, out_dir):
    from slamkit_amd.data import DataCollatorForLanguageModeling
    from slamkit_amd.trainer import SLAMTrainer, SLAMTrainingArguments
    args = SLAMTrainingArguments(output_dir=out_dir, per_device_train_batch_size=2, gradient_accumulation_steps=2,
                                 learning_rate=1e-2, warmup_steps=1, warmup_ratio=0.0, max_steps=3, logging_steps=1,
                                 ddp_bucket_layers=1, seed=5, save_steps=0, ddp_comm_dtype=comm, ddp_algo=algo)
    model = ShardStubLM()
    tr = SLAMTrainer(model=model, args=args, data_collator=DataCollatorForLanguageModeling(pad_token_id=0),
                     train_dataset=base.make_rows())
    tr.train()
    owned = list(getattr(tr.reducer, "owned", []))
    tr._gather_optimizer_state()
    return model, tr, owned


def _worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        for comm in ("float32", "bfloat16"):
            for algo in ("all_reduce", "rs_ag"):
                model, tr, owned = run(rank, world, algo, comm, os.path.join(tmp, f"r{rank}"))
                res[(comm, algo)] = (model.flat_params.tolist(), model.flat_master.tolist(), tr.exp_avg.tolist(), tr.exp_avg_sq.tolist(),
                                     [h["loss"] for h in tr.state.log_history], owned)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_rs_ag_equals_all_reduce_world2_gloo(tmp_path):
    world, port = 2, base._free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = base.V * base.H * 2
    for comm in ("float32", "bfloat16"):
        ar0, rs0, rs1 = res[0][(comm, "all_reduce")], res[0][(comm, "rs_ag")], res[1][(comm, "rs_ag")]
        # the sharded step really sharded: the two ranks own disjoint halves of every bucket, covering everything below the tail
        own0, own1 = rs0[5], rs1[5]
        assert own0 and own1 and not set(own0) & set(own1)
        assert sum(c for _, c in own0 + own1) == (n // (2 * CHUNK)) * 2 * CHUNK
        for i, name in enumerate(("params", "master", "exp_avg", "exp_avg_sq")):
            a, b, c = torch.tensor(rs0[i]), torch.tensor(rs1[i]), torch.tensor(ar0[i])
            assert torch.equal(a, b), (comm, name, "ranks differ")
            assert torch.equal(a, c), (comm, name, float((a - c).abs().max()))
        assert rs0[4] == ar0[4]  # logged losses
    # and the fp32 exchange reproduces the single-process run over the union of the batches (the base test's reference)
    ref_p, ref_losses, _ = base._single_process_reference(True)
    p = torch.tensor(res[0][("float32", "rs_ag")][0])
    assert torch.allclose(p, ref_p, rtol=1e-5, atol=1e-6), float((p - ref_p).abs().max())


# ---- DPO under rs_ag (SLAMDPOTrainer used the replicated clip + AdamW behind a sharded reducer) -------------
class DpoStubLM(ShardStubLM):
    """The UnitLM surface SLAMDPOTrainer drives: sequence_logps + backward_sequence_loss on the bag-of-embeddings LM."""

    class _Cfg:
        pad_token_id = 0
    config = _Cfg()

    def sequence_logps(self, input_ids, labels):
        p = self.flat_params.detach().clone().requires_grad_(True)
        E, W = p[: base.V * base.H].view(base.V, base.H), p[base.V * base.H:].view(base.V, base.H)
        logp = torch.log_softmax(E[input_ids] @ W.t(), dim=-1)[:, :-1]
        tgt = labels[:, 1:]
        m = tgt != -100
        ll = (logp.gather(-1, tgt.clamp(min=0)[..., None])[..., 0] * m).sum(1)
        self._graph = (ll, p)
        return ll.detach(), m.sum(1).float()

    def backward_sequence_loss(self, seq_coef, B, T, grad_scale=1.0, bucket_layers=0, bucket_cb=None):
        ll, p = self._graph
        (g,) = torch.autograd.grad((seq_coef * (-ll)).sum() * grad_scale, p)
        self.flat_grads.add_(g)
        if bucket_cb is not None:
            bucket_cb(base.V * base.H, base.V * base.H)
            bucket_cb(0, base.V * base.H)


def run_dpo(rank, world, algo, out_dir):
    from slamkit_amd.trainer import DPOConfig, SLAMDPOTrainer

    class Tok:
        bos_token_id = eos_token_id = 1
        def __call__(self, s, add_special_tokens=False):
            return {"input_ids": list(s)}
    g = torch.Generator().manual_seed(11)
    ids = lambda lo, hi: torch.randint(2, base.V, (int(torch.randint(lo, hi, (1,), generator=g)),), generator=g).tolist()  # noqa: E731
    rows = [{"prompt": ids(2, 5), "chosen": ids(3, 8), "rejected": ids(3, 8)} for _ in range(16)]
    args = DPOConfig(output_dir=out_dir, per_device_train_batch_size=2, gradient_accumulation_steps=2, learning_rate=5e-3,
                     warmup_steps=1, warmup_ratio=0.0, max_steps=2, logging_steps=1, ddp_bucket_layers=1, seed=5, save_steps=0,
                     ddp_comm_dtype="float32", ddp_algo=algo, beta=0.1)
    model, ref = DpoStubLM(seed=0), DpoStubLM(seed=0)
    tr = SLAMDPOTrainer(model=model, ref_model=ref, args=args, train_dataset=rows, processing_class=Tok())
    tr.train()
    return model, tr


def _dpo_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        for algo in ("all_reduce", "rs_ag"):
            model, tr = run_dpo(rank, world, algo, os.path.join(tmp, f"d{rank}"))
            # train() gathered the sharded optimizer state: master / moments are whole on every rank
            res[algo] = (model.flat_params.tolist(), model.flat_master.tolist(), tr.exp_avg.tolist(), type(tr.reducer).__name__)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_dpo_rs_ag_equals_all_reduce_world2_gloo(tmp_path):
    """SLAMDPOTrainer behind a ShardedGradReducer must run the sharded clip + AdamW and the parameter all-gather: both ranks
    end with the parameters (and, after train(), the gathered optimizer state) of the all_reduce run, bit for bit."""
    world, port = 2, base._free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dpo_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["rs_ag"][3] == "ShardedGradReducer" and res[0]["all_reduce"][3] == "GradBucketReducer"
    init = DpoStubLM(seed=0).flat_params
    assert not torch.equal(torch.tensor(res[0]["all_reduce"][0]), init)  # the steps moved the parameters
    for i, name in enumerate(("params", "master", "exp_avg")):
        a, b, c = torch.tensor(res[0]["rs_ag"][i]), torch.tensor(res[1]["rs_ag"][i]), torch.tensor(res[0]["all_reduce"][i])
        assert torch.equal(a, b), (name, "ranks differ")
        assert torch.equal(a, c), (name, float((a - c).abs().max()))
