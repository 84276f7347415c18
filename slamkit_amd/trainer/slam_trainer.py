"""SLAMTrainer on the HIP engine: the step loop transformers.Trainer runs for the reference
(/root/reference slamkit/trainer/slam_trainer.py:27-71 over site-packages transformers/trainer.py
_inner_training_loop) restated around UnitLM.forward/backward,
slam_grad_norm + slam_adamw_step and the bucketed RCCL reducer.

Differences from the reference that are deliberate (and loss-equivalent):
 * tokens-seen and num_items_in_batch are counted on the host from the collated CPU labels and
   all-reduced ONCE per optimizer step, instead of a device->host sync + all-gather on every
   micro-step (slam_trainer.py:70);
 * the loss is read back only every `logging_steps`;
 * gradients are all-reduced as a few large ranges of one flat fp32 buffer, overlapped with the
   backward of the last micro-batch (no_sync on the others, like trainer.py:1750-1758).
"""
from __future__ import annotations

import json
import logging
import math
import os
import shutil
import time
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from .callbacks import TrainerCallback, TrainerControl, TrainerState
from .dp import GradBucketReducer, ShardedGradReducer, all_reduce_scalar, host_group, seeded_batches, shard_batches, world_info
from .training_args import SLAMTrainingArguments, lr_lambda

logger = logging.getLogger(__name__)


class SLAMTrainer:
    def __init__(self, model=None, args: SLAMTrainingArguments = None, data_collator: Callable = None,
                 train_dataset=None, eval_dataset=None, processing_class=None, callbacks: Optional[List[TrainerCallback]] = None):
        self.model = model
        self.args = args or SLAMTrainingArguments()
        self.data_collator = data_collator
        self.train_dataset = train_dataset
        self.eval_dataset = eval_dataset
        self.processing_class = processing_class
        self.callbacks = list(callbacks or [])
        self.state = TrainerState()
        self.control = TrainerControl()
        self.rank, self.world = world_info()
        dev = model.device
        n = model.engine.n_params
        osd = getattr(self.args, "optim_state_dtype", "float32") or "float32"
        if osd not in ("float32", "bfloat16", "float32_bf16_moments"):
            raise ValueError(f"optim_state_dtype must be float32, bfloat16 or float32_bf16_moments, got {osd!r}")
        # state_dtype = precision of the WEIGHT state (fp32 master or the bf16 parameters themselves); moment_dtype = Adam moments
        self.state_dtype = torch.bfloat16 if osd == "bfloat16" else torch.float32
        self.moment_dtype = torch.float32 if osd == "float32" else torch.bfloat16
        if self.state_dtype == torch.bfloat16:
            if getattr(self.args, "overlap_optimizer", False):
                raise ValueError("overlap_optimizer is implemented for the fp32-state optimizer only")
            if hasattr(model, "drop_master"):
                model.drop_master()  # the bf16 parameters become the only copy (the recipe's torch_dtype: bfloat16)
        if self.moment_dtype == torch.bfloat16 and self.state_dtype == torch.float32 and getattr(self.args, "overlap_optimizer", False):
            raise ValueError("overlap_optimizer is implemented for the fp32-state optimizer only")
        gd = getattr(self.args, "grad_dtype", None) or ("bfloat16" if osd == "bfloat16" else "float32")
        if gd not in ("float32", "bfloat16"):
            raise ValueError(f"grad_dtype must be float32 or bfloat16, got {gd!r}")
        # how the last backward of a step delivers its final values (UnitLM.backward(final=)): 2 = bf16 only, 1 = fp32; both
        # emit the gradient-norm partials from the final-value stores. Models without the hook (CPU stubs of the gloo tests): 0
        self._final_mode = (2 if gd == "bfloat16" else 1) if hasattr(model, "enable_bf16_grads") else 0
        if self._final_mode:
            model.engine.set_option("grad_norm_partials", 1 if getattr(self.args, "grad_norm_from_backward", True) else 0)
        self.exp_avg = torch.zeros(n, dtype=self.moment_dtype, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=self.moment_dtype, device=dev)
        self.norm_out = torch.zeros(2, dtype=torch.float32, device=dev)
        cd = getattr(self.args, "ddp_comm_dtype", None)
        algo = getattr(self.args, "ddp_algo", "all_reduce") or "all_reduce"
        if algo not in ("all_reduce", "rs_ag"):
            raise ValueError(f"ddp_algo must be all_reduce or rs_ag, got {algo!r}")
        if algo == "rs_ag":
            if getattr(self.args, "overlap_optimizer", False):
                raise ValueError("ddp_algo=rs_ag runs the optimizer on shards; overlap_optimizer belongs to the replicated step")
            chunk, n_chunks = model.engine.grad_chunk_info()
            self.reducer = ShardedGradReducer(model.flat_grads, model.flat_params, chunk, comm_dtype=getattr(torch, cd) if cd else None,
                                              engine=model.engine)
            if args.logging_steps and hasattr(model.engine, "set_option"):
                try:  # the logged `exposed_param_gather_ms` needs the waits bracketed by timing events (two more packets per wait)
                    model.engine.set_option("time_param_waits", 1)
                except Exception:  # noqa: BLE001 - an engine without the option simply reports 0
                    pass
            self._chunk_sums = torch.zeros(n_chunks, dtype=torch.float32, device=dev)
        else:
            self.reducer = GradBucketReducer(model.flat_grads, comm_dtype=getattr(torch, cd) if cd else None, engine=model.engine)
        if self._final_mode == 2 and self.reducer.comm_dtype == torch.bfloat16:
            # bf16 final gradients + bf16 exchange: ONE bf16 buffer is the gradient store, the communication image and the
            # optimizer's input (training_step)
            self.reducer.stage = model.enable_bf16_grads()
        if (self.world > 1 or self.reducer.force) and torch.device(dev).type == "cuda":
            from .. import check_hw_queues
            check_hw_queues(8)
        self.host_group = host_group()  # None on a single rank or when gloo cannot be set up
        self._loss_acc = torch.zeros(1, dtype=torch.float32, device=dev)
        self._loss_n = 0
        self._loss_is_rank_mean = False  # subclasses whose per-rank loss is a local mean (DPO) set this
        self.opt_step = 0
        # engine option: AdamW + weight-image refresh in per-layer chunks on the engine's side stream; slam_forward
        # waits per layer. Everything that reads the flat buffers with torch goes through UnitLM (which joins first).
        model.engine.set_option("overlap_adamw", 1 if getattr(args, "overlap_optimizer", False) else 0)

    # ---- reference hooks ------------------------------------------------------------------------
    def get_num_tokens(self, labels: torch.Tensor) -> int:
        """slam_trainer.py:59-65 (host-side count on the collated labels)."""
        valid = labels != -100
        if self.args.min_token_id_count is not None:
            valid = torch.logical_and(valid, labels >= self.args.min_token_id_count)
        if self.args.max_token_id_count is not None:
            valid = torch.logical_and(valid, labels <= self.args.max_token_id_count)
        return int(valid.sum())

    def training_step(self, model, inputs: Dict[str, torch.Tensor], num_items_in_batch=None, last_micro: bool = True,
                      grad_scale: float = 1.0) -> torch.Tensor:
        """slam_trainer.py:67-71 + Trainer.training_step: forward + backward of one micro-batch.
        Returns the (device) loss tensor without synchronising."""
        out = model.forward(input_ids=inputs["input_ids"], attention_mask=inputs.get("attention_mask"),
                            position_ids=inputs.get("position_ids"), labels=inputs["labels"],
                            num_items_in_batch=num_items_in_batch, return_logits=False)
        loss = out.loss.detach()
        if last_micro and (self.world > 1 or self.reducer.force):
            # bf16 exchange: this backward writes the communication image itself (no pack pass); with bf16 final gradients the
            # image is ALSO where they live: backward skips the fp32 stores, the reduced values stay in the image (no widening
            # pass), the chunked norm and AdamW read them there
            keep = (self._final_mode == 2 and type(self).optimizer_step is SLAMTrainer.optimizer_step
                    and self.reducer.stage is not None and self.reducer.stage is getattr(model, "flat_grads16", None))
            armed = self.reducer.arm_image(keep_bf16=keep)
            if keep and armed:
                model.backward(grad_scale, self.args.ddp_bucket_layers, self.reducer.on_bucket, final=2)
            else:
                model.backward(grad_scale, self.args.ddp_bucket_layers, self.reducer.on_bucket)
        elif last_micro and self._final_mode:
            model.backward(grad_scale, final=self._final_mode)  # norm partials (and bf16-only final values) from this backward
        else:
            model.backward(grad_scale)
        return loss

    # ---- schedule -----------------------------------------------------------------------------------
    def _plan(self):
        a = self.args
        n = len(self.train_dataset)
        per_epoch_batches = math.ceil(n / a.per_device_train_batch_size)
        per_rank = math.ceil(per_epoch_batches / self.world)
        updates_per_epoch = max(1, math.ceil(per_rank / a.gradient_accumulation_steps))
        if a.max_steps and a.max_steps > 0:
            max_steps = a.max_steps
            epochs = math.ceil(max_steps / updates_per_epoch)
        else:
            max_steps = math.ceil(a.num_train_epochs * updates_per_epoch)
            epochs = math.ceil(a.num_train_epochs)
        return max_steps, epochs, updates_per_epoch

    def _epoch_batches(self, epoch: int):
        a = self.args
        batches = seeded_batches(len(self.train_dataset), a.per_device_train_batch_size, a.seed, epoch)
        return shard_batches(batches, self.rank, self.world, even=True)

    def _collate(self, idxs):
        return self.data_collator([self.train_dataset[i] for i in idxs])

    def _micro_batches(self, batches, ga: int):
        """Collated micro-batch groups of one epoch, one list per optimizer step. With dataloader_num_workers > 0
        (config/training_args/default.yaml:17 asks for 4 worker processes) ONE background thread collates up to two steps
        ahead into pinned host tensors, so the host side of step k+1 runs under the enqueue of step k and the H2D copies
        in UnitLM.forward are asynchronous. One thread is enough: collating a micro-batch of 8 x 1024 ids takes ~0.1 ms
        against a 25 ms step (bench.py `extras.host_boundary`: the whole host boundary costs 0.3 % without it). Order and
        contents are those of the synchronous path (tests/test_trainer_host.py)."""
        groups = [batches[s:s + ga] for s in range(0, len(batches), ga)]
        if self.args.dataloader_num_workers <= 0:
            for g in groups:
                yield [self._collate(b) for b in g]
            return
        import queue
        import threading
        pin = torch.cuda.is_available()
        q: "queue.Queue" = queue.Queue(maxsize=2)
        stop = threading.Event()

        def work():
            try:
                for g in groups:
                    if stop.is_set():
                        return
                    micro = [self._collate(b) for b in g]
                    if pin:
                        micro = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in m.items()} for m in micro]
                    q.put(micro)
                q.put(None)
            except BaseException as e:  # noqa: BLE001 - surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=work, name="slam-collate", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:  # the consumer stopped early (max_steps, stopper callbacks): release the producer
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)

    def _counts_ahead(self, groups):
        """(micro-batches of step k, handle of their count all-reduce), with the all-reduce of step k + 1 POSTED before step
        k is yielded - the blocking per-step collective of the step loop becomes a wait on work that finished a step ago.
        Subclasses with their own optimizer_step (DPO) keep their own counting: handle None."""
        if type(self).optimizer_step is not SLAMTrainer.optimizer_step or not (self.world > 1 or self.reducer.force):
            for micro in groups:
                yield micro, None
            return
        pending = None  # the step whose count collective is posted and not yet handed out
        try:
            for micro in groups:
                c = self.local_counts(micro)
                cur = (micro, (c, self.post_counts(*c)))
                if pending is not None:
                    ready, pending = pending, cur
                    yield ready
                else:
                    pending = cur
            if pending is not None:
                ready, pending = pending, None
                yield ready
        finally:
            # the consumer left early (max_steps, a stopper callback - decided identically on every rank, _sync_control): the
            # collective posted for the step that will not run is still in flight; drain it before save / evaluate /
            # destroy_process_group issue their own collectives on the group
            if pending is not None and pending[1][1][0] is not None:
                pending[1][1][0].wait()

    # ---- one optimizer step over `micro` collated CPU micro-batches -------------------------------------
    def local_counts(self, micro):
        """(num_items_in_batch, tokens seen) of this rank for one optimizer step, counted on the host from the CPU labels."""
        a = self.args
        local_items = float(sum(int((mb["labels"] != -100).sum()) for mb in micro))
        if a.min_token_id_count is None and a.max_token_id_count is None:
            return local_items, local_items
        return local_items, float(sum(self.get_num_tokens(mb["labels"]) for mb in micro))

    def post_counts(self, local_items: float, local_seen: float):
        """Start the all-reduce of one optimizer step's token counts and return a handle for `optimizer_step(counts_handle=)`.
        The reference gathers the counts with a device collective + `.item()` on EVERY micro-step (slam_trainer.py:70); here
        it is ONE host-side (gloo) all-reduce per optimizer step, and train() posts it one step AHEAD (the collate thread's
        batches are already there), so the step loop never waits for it. All ranks must post in the same order."""
        if not (self.world > 1 or self.reducer.force):
            return None, torch.tensor([local_items, local_seen], dtype=torch.float64)
        if self.host_group is not None:
            t = torch.tensor([local_items, local_seen], dtype=torch.float64)
            return dist.all_reduce(t, group=self.host_group, async_op=True), t
        # no gloo: device collective on the RCCL group (costs one host sync per step when the result is read)
        t = torch.tensor([local_items, local_seen], dtype=torch.float64, device=self.model.device)
        dist.all_reduce(t)
        return None, t

    def optimizer_step(self, micro: List[Dict[str, torch.Tensor]], lr: float, counts=None, counts_handle=None):
        """`counts` = (local num_items, local tokens seen) when the caller already knows them (device-
        resident synthetic batches); otherwise counted on the host from the CPU labels. `counts_handle` = what
        post_counts returned for THIS step's local counts (posted earlier); without it the all-reduce is posted here."""
        a = self.args
        if counts is not None:
            local_items, local_seen = float(counts[0]), float(counts[1])
        else:
            local_items, local_seen = self.local_counts(micro)
        if counts_handle is None:
            counts_handle = self.post_counts(local_items, local_seen)
        work, t = counts_handle
        if work is not None:
            work.wait()
        glob_items, glob_seen = (float(x) for x in t.tolist())
        if a.average_tokens_across_devices:
            n_items, scale = glob_items, 1.0           # sum over ranks of d(local_sum / global_count)
        else:
            n_items, scale = local_items, 1.0 / self.world  # per-rank token mean, then rank average
        for i, mb in enumerate(micro):
            if i == 0 and a.overwrite_first_grad:  # the first backward of the step stores the gradients: no zeroing pass
                self.model.engine.set_option("grad_overwrite_next", 1)
            loss = self.training_step(self.model, mb, num_items_in_batch=n_items, last_micro=(i == len(micro) - 1),
                                      grad_scale=scale)
            # every micro-batch loss is already normalised by the whole step's token count (global, or this rank's
            # when average_tokens_across_devices is off): their plain sum is the step's loss
            self._loss_acc += loss
        self._loss_n += 1
        self.reducer.finish()
        self.state.num_input_tokens_seen += int(glob_seen)
        self._update(lr, zero_grad=not a.overwrite_first_grad)
        self.state.global_step += 1

    def _update(self, lr: float, zero_grad: bool):
        """Clip + AdamW in the form the gradient exchange left the buffer in: a sharded reducer (ddp_algo = rs_ag) holds the
        summed gradients of this rank's shards only, so the update must be the sharded one followed by the parameter
        all-gather; every subclass step (DPO) goes through here too."""
        if getattr(self.reducer, "owned", None) is not None and (self.world > 1 or self.reducer.force):
            self._clip_and_update_sharded(lr, zero_grad=zero_grad)
        else:
            self._clip_and_update(lr, zero_grad=zero_grad)

    def _clip_and_update(self, lr: float, zero_grad: bool):
        """clip_grad_norm_ + AdamW on the flat buffers, in the configured state precision."""
        a, eng = self.args, self.model.engine
        eng.grad_norm(a.max_grad_norm if a.max_grad_norm else 0.0, self.norm_out)
        self.opt_step += 1
        if self.state_dtype == torch.bfloat16:
            eng.adamw_step_bf16(self.exp_avg, self.exp_avg_sq, self.norm_out, lr, a.adam_beta1, a.adam_beta2,
                                a.adam_epsilon, a.weight_decay, self.opt_step, zero_grad=zero_grad)
        else:
            eng.adamw_step(self.model.flat_master, self.exp_avg, self.exp_avg_sq, self.norm_out, lr, a.adam_beta1,
                           a.adam_beta2, a.adam_epsilon, a.weight_decay, self.opt_step, zero_grad=zero_grad)

    def _clip_and_update_sharded(self, lr: float, zero_grad: bool):
        """ddp_algo = rs_ag: after the reduce-scatters every rank holds the summed gradient of its own shard of each bucket
        (and of the small replicated tail). Global norm = chunk sums of the own shards, summed over the ranks (disjoint
        support: exact, and bit-identical to slam_grad_norm on the replicated buffer); AdamW on the owned ranges; bf16
        parameters all-gathered on the communication stream while the next forward starts."""
        a, eng, red = self.args, self.model.engine, self.reducer
        cs = self._chunk_sums
        cs.zero_()
        tail_off, tail_cnt = red.tail
        for off, cnt in red.owned:
            eng.grad_sumsq_chunks(off, cnt, cs)
        if tail_cnt and self.rank == 0:
            eng.grad_sumsq_chunks(tail_off, tail_cnt, cs)
        if self.world > 1 or red.force:
            dist.all_reduce(cs, op=dist.ReduceOp.SUM, group=red.group)
        eng.grad_norm_from_chunks(cs, a.max_grad_norm if a.max_grad_norm else 0.0, self.norm_out)
        self.opt_step += 1
        master = None if self.state_dtype == torch.bfloat16 else self.model.flat_master
        for off, cnt in list(red.owned) + ([(tail_off, tail_cnt)] if tail_cnt else []):
            eng.adamw_range(off, cnt, master, self.exp_avg, self.exp_avg_sq, self.norm_out, lr, a.adam_beta1, a.adam_beta2,
                            a.adam_epsilon, a.weight_decay, self.opt_step, zero_grad=False)
        if zero_grad:  # accumulate-mode backward: the regions this rank does not own hold partial sums, clear everything
            eng.zero_grads()
        red.gather_params(eng)
        self._shards_stale = True  # master / moments of the other ranks' shards are out of date until gathered (checkpoints)

    def _gather_optimizer_state(self):
        """Before a checkpoint under ddp_algo = rs_ag: bring the full-size master / moment buffers up to date from the
        owners of each shard (the file layout stays the replicated one, so a run may resume on any world size)."""
        red = self.reducer
        if not getattr(self, "_shards_stale", False) or not isinstance(red, ShardedGradReducer) or self.world == 1:
            return
        self.model.engine.join()
        w, r = self.world, self.rank
        bufs = [self.exp_avg, self.exp_avg_sq] + ([self.model.flat_master] if self.state_dtype == torch.float32 else [])
        for lo, hi in red.last_buckets:
            s = (hi - lo) // w
            for b in bufs:
                dist.all_gather_into_tensor(b[lo:hi], b[lo + r * s: lo + (r + 1) * s].clone(), group=red.group)
        self._shards_stale = False

    def _sync_control(self):
        """Callbacks decide from rank-local clocks (RunTimeStopperCallback: `start_time` differs per rank), and
        evaluate() / save_checkpoint() contain collectives: every rank must take the SAME stop / evaluate / save
        decision at the same step. One MAX all-reduce of the three flags, on the host (gloo) group when there is one
        so that the step loop does not synchronise with the device."""
        c = self.control
        if self.host_group is not None:
            t = torch.tensor([float(c.should_training_stop), float(c.should_evaluate), float(c.should_save)])
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.host_group)
        else:
            t = torch.tensor([float(c.should_training_stop), float(c.should_evaluate), float(c.should_save)],
                             device=self.model.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t = t.cpu()
        c.should_training_stop, c.should_evaluate, c.should_save = (bool(x > 0) for x in t.tolist())

    def _log(self, lr: float, t0: float, tokens0: int):
        loss_local = float(self._loss_acc) / max(1, self._loss_n)
        loss = all_reduce_scalar(loss_local, device=self.model.device) if self.world > 1 else loss_local
        if (not self.args.average_tokens_across_devices or self._loss_is_rank_mean) and self.world > 1:
            loss /= self.world  # each rank accumulated its own mean: report the mean over ranks
        self._loss_acc.zero_()
        self._loss_n = 0
        dt = time.time() - t0
        rec = {"step": self.state.global_step, "loss": loss, "grad_norm": float(self.norm_out[0]), "learning_rate": lr,
               "exposed_comm_ms": self.reducer.exposed_ms(),
               "exposed_param_gather_ms": (self.model.engine.param_wait_ms() if hasattr(self.model.engine, "param_wait_ms") else 0.0),
               "param_waits_untimed": (self.model.engine.param_wait_untimed() if hasattr(self.model.engine, "param_wait_untimed") else 0),
               "num_input_tokens_seen": self.state.num_input_tokens_seen,
               "tokens_per_sec": (self.state.num_input_tokens_seen - tokens0) / max(dt, 1e-9)}
        self.state.log_history.append(rec)
        if self.rank == 0:
            logger.info(json.dumps(rec))
        return rec

    # ---- train ------------------------------------------------------------------------------------------------
    def train(self, resume_from_checkpoint=None):
        a = self.args
        max_steps, epochs, updates_per_epoch = self._plan()
        self.state.max_steps = max_steps
        if resume_from_checkpoint:
            path = resume_from_checkpoint if isinstance(resume_from_checkpoint, str) else self._last_checkpoint()
            if path:
                self._load_checkpoint(path)
        self.model.zero_grad()
        for cb in self.callbacks:
            cb.on_train_begin(a, self.state, self.control)
        t0, tokens0 = time.time(), self.state.num_input_tokens_seen
        start_epoch = self.state.global_step // updates_per_epoch
        skip = (self.state.global_step % updates_per_epoch) * a.gradient_accumulation_steps
        done = self.state.global_step >= max_steps
        for epoch in range(start_epoch, max(epochs, start_epoch + 1)):
            if done:
                break
            batches = self._epoch_batches(epoch)[skip:]
            skip = 0
            steps = self._counts_ahead(self._micro_batches(batches, a.gradient_accumulation_steps))
            try:
                for micro, handle in steps:
                    lr = a.learning_rate * lr_lambda(a, self.state.global_step, max_steps)
                    if handle is None:
                        self.optimizer_step(micro, lr)
                    else:
                        self.optimizer_step(micro, lr, counts=handle[0], counts_handle=handle[1])
                    self.state.epoch = self.state.global_step / updates_per_epoch
                    for cb in self.callbacks:
                        cb.on_step_end(a, self.state, self.control)
                    if self.world > 1 and self.callbacks:
                        self._sync_control()
                    if self.state.global_step >= max_steps or self.control.should_training_stop:
                        done = True
                        steps.close()  # drains the count collective posted one step ahead BEFORE the collectives of log / evaluate / save
                    if a.logging_steps and self.state.global_step % a.logging_steps == 0:
                        self._log(lr, t0, tokens0)
                    if (a.eval_strategy == "steps" and a.eval_steps and self.state.global_step % a.eval_steps == 0) \
                            or self.control.should_evaluate:
                        self.evaluate()
                        self.control.should_evaluate = False
                    if (a.save_steps and self.state.global_step % a.save_steps == 0) or self.control.should_save:
                        self.save_checkpoint()
                        self.control.should_save = False
                    if done:
                        break
            finally:
                steps.close()
        if self._loss_n:
            self._log(a.learning_rate * lr_lambda(a, max(self.state.global_step - 1, 0), max_steps), t0, tokens0)
        # rs_ag: master weights / moments of the shards other ranks own are stale until gathered - bring them up to date so
        # that whatever reads the model or the optimizer state after train() (state_dict(float32), a final save) sees one copy
        self._gather_optimizer_state()
        for cb in self.callbacks:
            cb.on_train_end(a, self.state, self.control)
        if torch.device(self.model.device).type == "cuda":
            torch.cuda.synchronize(self.model.device)
        self.model.engine.join()  # a pending overlapped optimizer step: order it before whatever the caller does next
        return self.state

    @torch.no_grad()
    def evaluate(self, dataset=None) -> Dict[str, float]:
        ds = dataset if dataset is not None else self.eval_dataset
        if ds is None or len(ds) == 0:
            return {}
        bs = self.args.per_device_eval_batch_size
        idx = list(range(len(ds)))
        batches = shard_batches([idx[i:i + bs] for i in range(0, len(idx), bs)], self.rank, self.world, even=False)
        tot, cnt = torch.zeros(1, dtype=torch.float64, device=self.model.device), 0.0
        for b in batches:
            mb = self.data_collator([ds[i] for i in b])
            n = float(((mb["labels"][:, 1:]) != -100).sum())
            if n == 0:
                continue
            out = self.model.forward(input_ids=mb["input_ids"], position_ids=mb.get("position_ids"), labels=mb["labels"],
                                     num_items_in_batch=1.0, return_logits=False)
            tot += out.loss.double()
            cnt += n
        if self.world > 1:
            t = torch.cat([tot, torch.tensor([cnt], dtype=torch.float64, device=self.model.device)])
            dist.all_reduce(t)
            tot, cnt = t[:1], float(t[1])
        res = {"eval_loss": float(tot) / max(cnt, 1.0), "step": self.state.global_step}
        self.state.log_history.append(res)
        if self.rank == 0:
            logger.info(json.dumps(res))
        return res

    # ---- checkpoints --------------------------------------------------------------------------------------------
    def _ckpt_dir(self, step):
        return os.path.join(self.args.output_dir, f"checkpoint-{step}")

    def _last_checkpoint(self):
        d = self.args.output_dir
        if not os.path.isdir(d):
            return None
        c = [x for x in os.listdir(d) if x.startswith("checkpoint-") and x.split("-")[1].isdigit()]
        return os.path.join(d, max(c, key=lambda x: int(x.split("-")[1]))) if c else None

    def save_checkpoint(self):
        """HF-layout weights (UnitLM.save_pretrained) + optimizer/trainer state; keeps save_total_limit."""
        self._gather_optimizer_state()
        if self.rank == 0:
            path = self._ckpt_dir(self.state.global_step)
            self.model.save_pretrained(path)
            self.model.engine.join()
            torch.save({"master": self.model._weights.cpu(), "exp_avg": self.exp_avg.cpu(),
                        "exp_avg_sq": self.exp_avg_sq.cpu(), "opt_step": self.opt_step}, os.path.join(path, "optimizer.pt"))
            with open(os.path.join(path, "trainer_state.json"), "w") as f:
                json.dump({"global_step": self.state.global_step, "epoch": self.state.epoch,
                           "num_input_tokens_seen": self.state.num_input_tokens_seen,
                           "log_history": self.state.log_history}, f)
            if self.processing_class is not None and hasattr(self.processing_class, "save_pretrained"):
                self.processing_class.save_pretrained(path)
            lim = self.args.save_total_limit
            if lim:
                c = sorted([x for x in os.listdir(self.args.output_dir) if x.startswith("checkpoint-")],
                           key=lambda x: int(x.split("-")[1]))
                for old in c[:-lim]:
                    shutil.rmtree(os.path.join(self.args.output_dir, old), ignore_errors=True)
        if self.world > 1:
            dist.barrier()

    def _load_checkpoint(self, path: str):
        st = torch.load(os.path.join(path, "optimizer.pt"), map_location="cpu")
        self.model.engine.join()
        self.model._weights.copy_(st["master"])
        self.model.sync_params_from_master()
        self.exp_avg.copy_(st["exp_avg"])
        self.exp_avg_sq.copy_(st["exp_avg_sq"])
        self.opt_step = int(st["opt_step"])
        with open(os.path.join(path, "trainer_state.json")) as f:
            s = json.load(f)
        self.state.global_step = s["global_step"]
        self.state.epoch = s["epoch"]
        self.state.num_input_tokens_seen = s["num_input_tokens_seen"]
        self.state.log_history = s.get("log_history", [])
