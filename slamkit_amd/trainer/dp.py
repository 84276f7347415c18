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
    def __init__(self, flat_grads: torch.Tensor, group=None, comm_dtype: Optional[torch.dtype] = None, engine=None):
        """comm_dtype = torch.bfloat16 exchanges the buckets in bf16 through a staging buffer, like the reference's
        DDP does when the parameters (hence the gradients) are bf16: half the bytes on the xGMI links for two
        conversion passes per bucket; None (default) reduces the fp32 buffer in place.
        engine: the slamkit_amd Engine that owns `flat_grads` - the bf16 staging passes run as ITS kernels on the
        communication stream (slam_pack_grads_bf16 / slam_unpack_grads_bf16), so nothing between backward and the wire is a
        torch kernel; without one (the CPU stub models of the gloo tests) the conversion is a tensor copy."""
        self.flat = flat_grads
        self.engine = engine if (engine is not None and hasattr(engine, "pack_grads_bf16")) else None
        self.group = group
        self.comm_dtype = comm_dtype if comm_dtype not in (None, flat_grads.dtype) else None
        self.stage = None
        self.rank, self.world = world_info()
        self.pending = []
        self.ranges = []
        self.side = torch.cuda.Stream(device=flat_grads.device) if flat_grads.is_cuda else None
        # SLAM_DP_FORCE=1: run the collective path even on a single rank (exercises RCCL on a 1-GPU box)
        self.force = os.environ.get("SLAM_DP_FORCE", "0") == "1" and dist.is_available() and dist.is_initialized()
        self.time_buckets = False   # bracket every bucket's collective with timing events on the communication stream
        self._bucket_ev, self._last_bucket_ev = [], []
        # round 6: the step's final gradients may LIVE in the bf16 image (engine option "grad_final_next" = 2, the reference's own
        # gradient precision): then the reduced values stay in it - no widening pass; the norm and AdamW read them there
        self.keep_bf16 = False

    def arm_image(self, keep_bf16: bool = False):
        """Before the backward whose buckets will be exchanged (the last micro-batch of a step): with a bf16 exchange and an
        engine that can do it, backward itself writes the bf16 communication image of the gradients (slam_set_grad_image) and
        the per-bucket pack pass is skipped for this step. keep_bf16: the caller runs that backward with final = 2 - the image
        then holds the ONLY copy of the step's gradients and the exchanged values stay there (no widening pass either).
        Returns whether the image is armed."""
        self._image = False
        self.keep_bf16 = False
        if (self.engine is None or not hasattr(self.engine, "set_grad_image") or self.comm_dtype != torch.bfloat16
                or not self.flat.is_cuda or (self.world == 1 and not self.force) or os.environ.get("SLAM_DP_NO_IMAGE", "0") == "1"):
            return
        if self.stage is None:
            self.stage = torch.empty(self.flat.numel(), dtype=self.comm_dtype, device=self.flat.device)
        self.engine.set_grad_image(self.stage)
        self._image = True
        self.keep_bf16 = bool(keep_bf16)
        return True

    def _pack(self, offset: int, count: int, st: torch.Tensor):
        """st[0:count] = comm_dtype(grads[offset:offset+count]) on the current stream."""
        if getattr(self, "_image", False):
            return  # backward wrote the image of this range already
        if self.engine is not None and self.flat.is_cuda and st.dtype == torch.bfloat16 and not ((offset | count) & 3):
            self.engine.pack_grads_bf16(offset, count, st)
        else:
            st.copy_(self.flat[offset:offset + count])

    def _unpack(self, offset: int, count: int, st: torch.Tensor):
        """grads[offset:offset+count] = fp32(st[0:count]) on the current stream."""
        if self.keep_bf16:
            return  # the optimizer reads the reduced values in the image
        if self.engine is not None and self.flat.is_cuda and st.dtype == torch.bfloat16 and not ((offset | count) & 3):
            self.engine.unpack_grads_bf16(offset, count, st)
        else:
            self.flat[offset:offset + count].copy_(st)

    def _timed(self, tag, fn):
        """Run fn() (collectives on the current = communication stream); with time_buckets, between two timing events."""
        if not self.time_buckets or self.side is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(self.side)
        r = fn()
        e1.record(self.side)
        self._bucket_ev.append((tag, e0, e1))
        return r

    def bucket_ms(self):
        """[(what, offset, count, ms on the communication stream)] of the last finished step (time_buckets = True): the
        duration of each bucket's collective(s) where they ran - beside backward - not their exposed part."""
        out = []
        for (what, off, cnt), e0, e1 in self._last_bucket_ev:
            e1.synchronize()
            out.append((what, int(off), int(cnt), round(float(e0.elapsed_time(e1)), 3)))
        return out

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
            def exchange():
                if self.comm_dtype is None:
                    self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                else:  # cast -> reduce -> cast back, all ordered on the side stream
                    st = self.stage[offset:offset + count]
                    self._pack(offset, count, st)
                    dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
                    self._unpack(offset, count, st)
            with torch.cuda.stream(self.side):
                self._timed(("all_reduce", offset, count), exchange)
        elif self.comm_dtype is None:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            st = self.stage[offset:offset + count]
            self._pack(offset, count, st)
            dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
            self._unpack(offset, count, st)

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
        self._image = False
        self.keep_bf16 = False
        self._last_bucket_ev, self._bucket_ev = self._bucket_ev, []
        covered = sorted(self.ranges)
        self.ranges = []
        return covered


class ShardedGradReducer(GradBucketReducer):
    """ddp_algo = "rs_ag": the exchange SURVEY.md section 5 / 8e names for 8 GPUs on 7 point-to-point xGMI links each.

    Per gradient bucket the ranks REDUCE-SCATTER the gradients (every rank ends up with the summed 1/N shard it owns:
    each of its 7 links carries one shard concurrently), the optimizer then runs on the owned shards only (AdamW time
    and optimizer-state traffic / N), and the updated bf16 PARAMETERS are ALL-GATHERED bucket by bucket in the order
    the next forward needs them - the engine waits for each bucket right before its first read (slam_add_param_wait),
    so the gather of the later layers runs under the first layers' kernels. Same bytes on the wire as a bf16
    all-reduce of the gradients, no fp32 -> bf16 -> fp32 round trip of the whole buffer for the second half.

    Buckets are the engine's reported ranges re-cut at multiples of world x grad-norm-chunk elements, so that every
    shard is a whole number of norm chunks (the global gradient norm is then bit-identical to the replicated step:
    include/slam_engine.h, slam_grad_sumsq_chunks). The few thousand elements above the last multiple (the top of the
    flat buffer) stay replicated: all-reduced and updated by every rank."""

    def __init__(self, flat_grads, flat_params, chunk_elems: int, group=None, comm_dtype=None, engine=None):
        super().__init__(flat_grads, group=group, comm_dtype=comm_dtype, engine=engine)
        self.params = flat_params
        self.n = flat_grads.numel()
        w = max(1, self.world)
        self.align = w * int(chunk_elems)
        self.top = (self.n // self.align) * self.align   # [top, n): replicated tail
        self.cut_hi = self.top
        self.buckets = []   # (lo, hi) of this step's communicated buckets
        self.owned = []     # (offset, count) this rank owns after finish()
        self.active = False
        self._tail_done = False   # the replicated tail [top, n) of this step has been handed to the communication stream
        self._ag_events = []

    @property
    def tail(self):
        return (self.top, self.n - self.top)

    def _on_side(self, ready_stream, fn, tag=None):
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.ExternalStream(ready_stream, device=self.flat.device) if ready_stream
                      else torch.cuda.current_stream(self.flat.device))
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                self._timed(tag, fn) if tag is not None else fn()
        else:
            fn()

    def _reduce_scatter(self, lo: int, hi: int):
        w, r = max(1, self.world), self.rank
        s = (hi - lo) // w
        view = self.flat[lo:hi]
        if self.comm_dtype is None:
            dist.reduce_scatter_tensor(view[r * s:(r + 1) * s], view, op=dist.ReduceOp.SUM, group=self.group)
        else:  # cast the bucket, exchange in bf16, cast the owned shard back
            if self.stage is None:
                self.stage = torch.empty(self.n, dtype=self.comm_dtype, device=self.flat.device)
            st = self.stage[lo:hi]
            self._pack(lo, hi - lo, st)
            dist.reduce_scatter_tensor(st[r * s:(r + 1) * s], st, op=dist.ReduceOp.SUM, group=self.group)
            self._unpack(lo + r * s, s, st[r * s:(r + 1) * s])

    def on_bucket(self, offset: int, count: int, ready_stream: Optional[int] = None):
        self.ranges.append((offset, count))
        if (self.world == 1 and not self.force) or count <= 0:
            return
        self.active = True
        if not self._tail_done and self.top < self.n and offset <= self.top:
            # the replicated tail [top, n) is final once a reported range reaches down to `top` - normally the first one; a
            # model smaller than world x chunk (top == 0) or a first range shorter than the tail gets it with a later callback
            self._tail_done = True
            t0 = self.top

            def tail():  # in the precision of the buckets: every gradient crosses the wire the same way under both algorithms
                if self.comm_dtype is None:
                    dist.all_reduce(self.flat[t0:], op=dist.ReduceOp.SUM, group=self.group)
                else:
                    if self.stage is None:
                        self.stage = torch.empty(self.n, dtype=self.comm_dtype, device=self.flat.device)
                    st = self.stage[t0:]
                    self._pack(t0, self.n - t0, st)
                    dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
                    self._unpack(t0, self.n - t0, st)
            self._on_side(ready_stream, tail, tag=("all_reduce (replicated tail)", t0, self.n - t0))
        lo = 0 if offset == 0 else min(self.cut_hi, -(-offset // self.align) * self.align)
        if lo < self.cut_hi:
            hi = self.cut_hi
            self._on_side(ready_stream, lambda: self._reduce_scatter(lo, hi), tag=("reduce_scatter", lo, hi - lo))
            self.buckets.append((lo, hi))
            self.cut_hi = lo

    def finish(self):
        covered = super().finish() if not self.active else self._finish_active()
        return covered

    def _finish_active(self):
        if self.side is not None:
            cur = torch.cuda.current_stream(self.flat.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            cur.wait_stream(self.side)
            e1.record(cur)
            self._exposed = (e0, e1)
        assert self.cut_hi == 0, "backward did not report the range down to offset 0"
        w, r = max(1, self.world), self.rank
        self.owned = [(lo + r * ((hi - lo) // w), (hi - lo) // w) for lo, hi in self.buckets]
        self.last_buckets = list(self.buckets)
        self._last_bucket_ev, self._bucket_ev = self._bucket_ev, []
        assert self._tail_done or self.top >= self.n, "the replicated tail was never exchanged"
        self.buckets, self.cut_hi, self.active, self._tail_done = [], self.top, False, False
        self._image = False
        self.keep_bf16 = False
        covered = sorted(self.ranges)
        self.ranges = []
        return covered

    def gather_params(self, engine):
        """All-gather the updated bf16 parameters, lowest offsets (the layers the next forward reads first) first, on
        the communication stream; each bucket is handed to the engine as a pending parameter write."""
        w, r = max(1, self.world), self.rank
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.flat.device))
        self._ag_events = []
        for lo, hi in sorted(self.last_buckets):
            s = (hi - lo) // w
            out, inp = self.params[lo:hi], self.params[lo + r * s: lo + (r + 1) * s]
            if self.side is not None:
                with torch.cuda.stream(self.side):
                    if self.time_buckets:  # reported with the NEXT step's buckets (the gather runs under its forward)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(self.side)
                        dist.all_gather_into_tensor(out, inp, group=self.group)
                        e1.record(self.side)
                        self._bucket_ev.append((("all_gather (parameters)", lo, hi - lo), e0, e1))
                    else:
                        dist.all_gather_into_tensor(out, inp, group=self.group)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                engine.add_param_wait(lo, hi - lo, ev)
                self._ag_events.append(ev)
            else:
                dist.all_gather_into_tensor(out, inp, group=self.group)


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
