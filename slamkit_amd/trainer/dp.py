"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" on CPU for tests). Replaces torch DDP's reducer:

 * gradients live in ONE flat fp32 buffer; the engine reports contiguous ranges as they become
   final during backward (slam_bucket_cb) and each range is all-reduced (SUM) on a side stream
   while backward continues - large few buckets, which suits xGMI's point-to-point links better
   than DDP's 25 MB default;
 * the loss is normalised by the GLOBAL token count (all-reduced once per optimizer step), so the
   summed gradient is the exact global token-mean gradient and no averaging pass is needed;
 * batches are dealt to ranks round-robin like accelerate's BatchSamplerShard
   (accelerate/data_loader.py:113-150, split_batches=False).
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional, Sequence

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


_HOST_GROUP = None


def host_group():
    """A gloo group for host-side scalars (token counts, stop flags): keeps tiny collectives off the
    GPU streams. All ranks must call this at the same point (SLAMTrainer.__init__)."""
    global _HOST_GROUP
    force = os.environ.get("SLAM_DP_FORCE", "0") == "1"
    if _HOST_GROUP is None and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        if dist.get_backend() == "gloo":
            _HOST_GROUP = dist.group.WORLD
        else:
            # single-node rendezvous on 127.0.0.1: pin gloo to loopback (the container hostname may not resolve)
            if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            try:
                _HOST_GROUP = dist.new_group(backend="gloo")
            except Exception as e:  # noqa: BLE001 - fall back to device-side scalars on the default group
                import logging
                logging.getLogger(__name__).warning(f"gloo side group unavailable ({e}); using the RCCL group for scalars")
                _HOST_GROUP = False
    return _HOST_GROUP or None


class GradBucketReducer:
    def __init__(self, flat_grads: torch.Tensor, group=None, comm_dtype: Optional[torch.dtype] = None):
        """comm_dtype = torch.bfloat16 exchanges the buckets in bf16 through a staging buffer, like the reference's
        DDP does when the parameters (hence the gradients) are bf16: half the bytes on the xGMI links for two
        conversion passes per bucket; None (default) reduces the fp32 buffer in place."""
        self.flat = flat_grads
        self.group = group
        self.comm_dtype = comm_dtype if comm_dtype not in (None, flat_grads.dtype) else None
        self.stage = None
        self.rank, self.world = world_info()
        self.pending = []
        self.ranges = []
        self.side = torch.cuda.Stream(device=flat_grads.device) if flat_grads.is_cuda else None
        # SLAM_DP_FORCE=1: run the collective path even on a single rank (exercises RCCL on a 1-GPU box)
        self.force = os.environ.get("SLAM_DP_FORCE", "0") == "1" and dist.is_available() and dist.is_initialized()

    def exposed_ms(self) -> float:
        """Exposed gradient-exchange time of the last finished step (synchronises on its end event); 0 on one rank."""
        ev = getattr(self, "_exposed", None)
        if ev is None:
            return 0.0
        ev[1].synchronize()
        return float(ev[0].elapsed_time(ev[1]))

    def on_bucket(self, offset: int, count: int, ready_stream: Optional[int] = None):
        """Called by the engine (host side) right after the kernels producing grads[offset:offset+count]
        were enqueued. ready_stream: raw handle of the stream the range is complete on (slam_bucket_stream: the engine's
        weight-gradient stream for the intermediate buckets); None = the current stream."""
        self.ranges.append((offset, count))
        if (self.world == 1 and not self.force) or count <= 0:
            return
        view = self.flat[offset:offset + count]
        if self.comm_dtype is not None and self.stage is None:
            self.stage = torch.empty(self.flat.numel(), dtype=self.comm_dtype, device=self.flat.device)
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.ExternalStream(ready_stream, device=self.flat.device) if ready_stream
                      else torch.cuda.current_stream(self.flat.device))
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                if self.comm_dtype is None:
                    self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                else:  # cast -> reduce -> cast back, all ordered on the side stream
                    st = self.stage[offset:offset + count]
                    st.copy_(view)
                    dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
                    view.copy_(st)
        elif self.comm_dtype is None:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            st = self.stage[offset:offset + count]
            st.copy_(view)
            dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(st)

    def finish(self):
        """Make the compute stream wait for every outstanding bucket. The stall of the compute stream (= the part of
        the gradient exchange that backward did not hide) is bracketed by two events; `exposed_ms()` reads it."""
        if self.side is not None and (self.pending or self.comm_dtype is not None) and self.ranges:
            cur = torch.cuda.current_stream(self.flat.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for w in self.pending:
                w.wait()
            cur.wait_stream(self.side)
            e1.record(cur)
            self._exposed = (e0, e1)
        else:
            for w in self.pending:
                w.wait()
        self.pending = []
        covered = sorted(self.ranges)
        self.ranges = []
        return covered


def all_reduce_scalar(value: float, device=None, group=None, dtype=torch.float64) -> float:
    rank, world = world_info()
    if world == 1:
        return value
    t = torch.tensor([value], dtype=dtype, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) if dtype.is_floating_point else int(t.item())


def shard_batches(batches: Sequence, rank: int, world: int, even: bool = True) -> List:
    """Rank r takes batches r, r+world, ... ; with `even`, the tail is completed by cycling from
    the start so every rank runs the same number of steps (BatchSamplerShard even_batches=True)."""
    n = len(batches)
    if world == 1:
        return list(batches)
    full = (n // world) * world
    mine = [batches[i] for i in range(rank, full, world)]
    rem = n - full
    if rem and even:
        tail = list(batches[full:]) + list(batches[: world - rem])
        mine.append(tail[rank])
    elif rem and rank < rem:
        mine.append(batches[full + rank])
    return mine


def seeded_batches(num_samples: int, batch_size: int, seed: int, epoch: int, drop_last: bool = False) -> List[List[int]]:
    """Seeded shuffle -> consecutive per-device batches (RandomSampler + BatchSampler)."""
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(num_samples, generator=g).tolist()
    out = [perm[i:i + batch_size] for i in range(0, num_samples, batch_size)]
    if drop_last and out and len(out[-1]) < batch_size:
        out.pop()
    return out
