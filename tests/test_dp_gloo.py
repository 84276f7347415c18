"""Multi-process CPU tests (gloo, world_size 2) of the data-parallel path: bucketed gradient
all-reduce, host-side token-count group, and the global-token-mean equivalence the trainer relies on
(sum over ranks of d(local_sum / global_count) == single-process gradient of the concatenated batch).
The compute inside the workers is the CPU oracle (test infrastructure) - the engine needs a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import slam_oracle as O
        from slamkit_amd.trainer.dp import GradBucketReducer, host_group, shard_batches, world_info
        assert world_info() == (rank, world)
        hg = host_group()
        # 1) host-side scalar all-reduce (token counts)
        t = torch.tensor([10.0 + rank, 1.0], dtype=torch.float64)
        dist.all_reduce(t, group=hg)
        assert t.tolist() == [21.0, 2.0]

        # 2) bucket reducer over a flat buffer, ranges reported in backward order
        n = 1000
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
        red = GradBucketReducer(flat)
        for off, cnt in [(700, 300), (300, 400), (0, 300)]:
            red.on_bucket(off, cnt)
        covered = red.finish()
        assert covered == [(0, 300), (300, 400), (700, 300)]
        assert torch.equal(flat, torch.arange(n, dtype=torch.float32) * 3)

        # 2b) the same exchange through a bf16 staging buffer (ddp_comm_dtype="bfloat16"): values here are exact in bf16
        base_b = (torch.arange(256) % 64).float()  # small integers: exact in bf16, also after the sum
        flat_b = base_b * (rank + 1)
        red_b = GradBucketReducer(flat_b, comm_dtype=torch.bfloat16)
        red_b.on_bucket(128, 128)
        red_b.on_bucket(0, 128)
        assert red_b.finish() == [(0, 128), (128, 128)]
        assert torch.equal(flat_b, base_b * 3)

        # 2c) sharded reducer (rs_ag): the replicated tail [top, n) is exchanged by the first callback that REACHES `top`,
        #     not blindly by the first one (a first range shorter than the tail: its gradients are not final yet), and a
        #     buffer smaller than world x chunk (top == 0) is all tail, exchanged with the last callback
        from slamkit_amd.trainer.dp import ShardedGradReducer
        n3, chunk = 200, 16                       # align = 32, top = 192, tail = [192, 200)
        g3 = torch.zeros(n3)
        red3 = ShardedGradReducer(g3, torch.zeros(n3), chunk)
        assert red3.tail == (192, 8)
        g3[196:] = rank + 1.0                      # "backward" wrote only the top of the buffer so far
        red3.on_bucket(196, 4)                     # shorter than the tail: nothing may be exchanged yet
        assert not red3._tail_done and not red3.buckets
        g3[:196] = rank + 1.0                      # ... the rest arrives
        red3.on_bucket(64, 132)
        assert red3._tail_done and red3.buckets == [(64, 192)]
        red3.on_bucket(0, 64)
        red3.finish()
        assert torch.equal(g3[192:], torch.full((8,), 3.0))          # whole tail reduced once, after it was final
        assert all(float(g3[o]) == 3.0 and float(g3[o + c - 1]) == 3.0 for o, c in red3.owned)
        small = torch.full((20,), rank + 1.0)      # n < align: top == 0
        red4 = ShardedGradReducer(small, torch.zeros(20), chunk)
        red4.on_bucket(10, 10)
        assert not red4._tail_done
        red4.on_bucket(0, 10)
        red4.finish()
        assert torch.equal(small, torch.full((20,), 3.0)) and red4.owned == []

        # 3) DP gradient equivalence with the oracle as the compute
        cfg = O.OracleConfig(n_layers=1, hidden=64, n_heads=1, n_kv_heads=1, head_dim=64, intermediate=128)
        sd = O.init_weights(cfg, seed=1, bias_std=0.02)
        g = torch.Generator().manual_seed(5)
        ids = torch.randint(2, 502, (4, 24), generator=g)
        labels = ids.clone()
        labels[1, 15:] = -100
        labels[3, 9:] = -100
        batches = [[0, 1], [2, 3]]
        mine = shard_batches(batches, rank, world)[0]
        cnt = torch.tensor([float((labels[mine] != -100).sum())], dtype=torch.float64)
        dist.all_reduce(cnt, group=hg)
        n_glob = float(cnt)
        assert n_glob == float((labels != -100).sum())
        _, _, grads = O.forward_loss_grads(cfg, sd, ids[mine], labels[mine], num_items_in_batch=n_glob)
        keys = sorted(grads)
        flatg = torch.cat([grads[k].flatten() for k in keys])
        red2 = GradBucketReducer(flatg)
        half = flatg.numel() // 2
        red2.on_bucket(half, flatg.numel() - half)
        red2.on_bucket(0, half)
        red2.finish()
        _, _, ref = O.forward_loss_grads(cfg, sd, ids, labels, num_items_in_batch=n_glob)
        refg = torch.cat([ref[k].flatten() for k in keys])
        err = float((flatg - refg).norm() / refg.norm())
        q.put((rank, err))
    finally:
        dist.destroy_process_group()


def test_dp_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err in res:
        assert err < 1e-5, (rank, err)
