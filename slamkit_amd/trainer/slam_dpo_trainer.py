"""Preference optimisation (DPO) on the HIP engine: /root/reference cli/preference_alignment_train.py:18-65
+ slamkit/trainer/slam_dpo_trainer.py:4-64 (tokenize_row) + TRL's sigmoid DPO loss (TRL is absent here:
the loss is restated from its definition, SURVEY.md §8c - parity for it is unpinned, tokenize_row is pinned).

Policy and frozen reference model are both `slamkit_amd.model.UnitLM`; one optimizer step =
  policy forward on [chosen; rejected] (2B sequences) -> per-sequence completion log-probs (slam_seq_loglik)
  reference forward (no gradients)                   -> ref log-probs
  x = beta * ((pi_c - pi_r) - (ref_c - ref_r)); loss = mean(-log sigmoid(x))
  d loss / d(-logp) per sequence -> slam_scale_loss_rows -> slam_backward -> clip + AdamW
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .slam_trainer import SLAMTrainer
from .training_args import SLAMTrainingArguments


@dataclass
class DPOConfig(SLAMTrainingArguments):
    """config/training_args/dpo_training_args.yaml: lr 5e-5, beta 0.1."""
    beta: float = 0.1
    max_prompt_length: Optional[int] = 512
    max_completion_length: Optional[int] = None
    max_length: Optional[int] = 1024
    learning_rate: float = 5e-5


class SLAMDPOTrainer(SLAMTrainer):
    @staticmethod
    def tokenize_row(features, processing_class, max_prompt_length, max_completion_length, add_special_tokens=False):
        """slam_dpo_trainer.py:7-64: prompt = [bos] + ids (left-truncated), completions = ids + [eos]
        (right-truncated); strings are tokenised without special tokens."""
        tokenizer = processing_class  # the UnitTokeniser itself, as preference_alignment_train.py:56-58 passes it
        enc = lambda s: list(tokenizer(s, add_special_tokens=False)["input_ids"])  # noqa: E731
        prompt_input_ids = [tokenizer.bos_token_id] + enc(features["prompt"])
        if add_special_tokens and tokenizer.eos_token_id is not None:
            prompt_input_ids = prompt_input_ids + [tokenizer.eos_token_id]
        chosen_input_ids = enc(features["chosen"]) + [tokenizer.eos_token_id]
        rejected_input_ids = enc(features["rejected"]) + [tokenizer.eos_token_id]
        if max_prompt_length is not None:
            prompt_input_ids = prompt_input_ids[-max_prompt_length:]
        if max_completion_length is not None:
            chosen_input_ids = chosen_input_ids[:max_completion_length]
            rejected_input_ids = rejected_input_ids[:max_completion_length]
        return {"prompt_input_ids": prompt_input_ids, "chosen_input_ids": chosen_input_ids,
                "rejected_input_ids": rejected_input_ids}

    def __init__(self, model=None, ref_model=None, args: DPOConfig = None, train_dataset=None, eval_dataset=None,
                 processing_class=None, callbacks=None):
        args = args or DPOConfig()
        tok = lambda ds: None if ds is None else [  # noqa: E731
            self.tokenize_row(r, processing_class, args.max_prompt_length, args.max_completion_length) for r in ds]
        super().__init__(model=model, args=args, data_collator=self._collate_pairs, train_dataset=tok(train_dataset),
                         eval_dataset=tok(eval_dataset), processing_class=processing_class, callbacks=callbacks)
        self.ref_model = ref_model
        self.pad_id = model.config.pad_token_id
        self._loss_is_rank_mean = True  # each rank accumulates the mean DPO loss of its own pairs

    def _collate_pairs(self, rows: List[Dict[str, List[int]]]) -> Dict[str, torch.Tensor]:
        """[chosen rows; rejected rows], right-padded; labels = completion tokens only (prompt and pad -> -100)."""
        seqs, labs = [], []
        for key in ("chosen_input_ids", "rejected_input_ids"):
            for r in rows:
                ids = (r["prompt_input_ids"] + r[key])
                lab = [-100] * len(r["prompt_input_ids"]) + list(r[key])
                if self.args.max_length is not None:
                    ids, lab = ids[: self.args.max_length], lab[: self.args.max_length]
                seqs.append(ids)
                labs.append(lab)
        T = max(len(s) for s in seqs)
        T = -(-T // 64) * 64  # token count a multiple of 64: keeps the step on the LDS-DMA wgrad path (padding is masked)
        ids = torch.full((len(seqs), T), self.pad_id, dtype=torch.long)
        lab = torch.full((len(seqs), T), -100, dtype=torch.long)
        for i, (s, l) in enumerate(zip(seqs, labs)):
            ids[i, : len(s)] = torch.tensor(s)
            lab[i, : len(l)] = torch.tensor(l)
        return {"input_ids": ids, "labels": lab}

    @staticmethod
    def dpo_loss(pi_c, pi_r, ref_c, ref_r, beta: float):
        x = beta * ((pi_c - pi_r) - (ref_c - ref_r))
        return -F.logsigmoid(x), x

    def optimizer_step(self, micro, lr: float, counts=None):
        a = self.args
        nm = len(micro)
        for i, mb in enumerate(micro):
            ids, lab = mb["input_ids"], mb["labels"]
            B2, T = ids.shape
            n = B2 // 2
            with torch.no_grad():
                ref, _ = self.ref_model.sequence_logps(ids, lab)
                ref = ref.clone()
            pol, _ = self.model.sequence_logps(ids, lab)
            losses, x = self.dpo_loss(pol[:n], pol[n:], ref[:n], ref[n:], a.beta)
            # d mean(loss) / d(-logp): chosen +beta*sigmoid(-x)/n, rejected -beta*sigmoid(-x)/n
            g = a.beta * torch.sigmoid(-x) / (n * nm * self.world)
            coef = torch.cat([g, -g])
            last = i == nm - 1
            dp = last and (self.world > 1 or self.reducer.force)
            if dp:
                self.reducer.arm_image()
            self.model.backward_sequence_loss(coef, B2, T, 1.0, bucket_layers=a.ddp_bucket_layers if dp else 0,
                                              bucket_cb=self.reducer.on_bucket if dp else None)
            self._loss_acc += losses.mean().detach() / nm
        self._loss_n += 1
        seen = float(sum(int((mb["labels"] != -100).sum()) for mb in micro))
        if self.world > 1:  # the real count over ranks (host-side group; one tiny collective per optimizer step)
            t = torch.tensor([seen], dtype=torch.float64, device="cpu" if self.host_group is not None else self.model.device)
            torch.distributed.all_reduce(t, group=self.host_group)
            seen = float(t)
        self.state.num_input_tokens_seen += int(seen)
        self.reducer.finish()
        self._update(lr, zero_grad=True)  # replicated or sharded (ddp_algo = rs_ag), as the reducer left the gradients
        self.state.global_step += 1

    @torch.no_grad()
    def evaluate(self, dataset=None) -> Dict[str, float]:
        """DPO validation: mean sigmoid-DPO loss, reward accuracy (chosen reward > rejected reward) and reward margin over
        the tokenised validation pairs - what TRL's DPOTrainer reports as eval_loss / eval_rewards/accuracies /
        eval_rewards/margins (restated from the definitions; rewards = beta * (policy logp - reference logp))."""
        from .dp import shard_batches
        ds = dataset if dataset is not None else self.eval_dataset
        if ds is None or len(ds) == 0:
            return {}
        if len(ds) and "prompt_input_ids" not in ds[0]:
            ds = [self.tokenize_row(r, self.processing_class, self.args.max_prompt_length, self.args.max_completion_length)
                  for r in ds]
        bs = self.args.per_device_eval_batch_size
        idx = list(range(len(ds)))
        batches = shard_batches([idx[i:i + bs] for i in range(0, len(idx), bs)], self.rank, self.world, even=False)
        acc = torch.zeros(4, dtype=torch.float64, device=self.model.device)  # loss sum, correct, margin sum, pairs
        for b in batches:
            mb = self._collate_pairs([ds[i] for i in b])
            ids, lab = mb["input_ids"], mb["labels"]
            n = ids.shape[0] // 2
            ref, _ = self.ref_model.sequence_logps(ids, lab)
            ref = ref.clone()
            pol, _ = self.model.sequence_logps(ids, lab)
            losses, x = self.dpo_loss(pol[:n], pol[n:], ref[:n], ref[n:], self.args.beta)
            acc += torch.stack([losses.sum(), (x > 0).sum().to(losses.dtype), x.sum(), x.new_tensor(float(n))]).double()
        if self.world > 1:
            torch.distributed.all_reduce(acc)
        tot = acc.tolist()
        k = max(tot[3], 1.0)
        res = {"eval_loss": tot[0] / k, "eval_rewards/accuracies": tot[1] / k, "eval_rewards/margins": tot[2] / k,
               "step": self.state.global_step}
        self.state.log_history.append(res)
        return res
