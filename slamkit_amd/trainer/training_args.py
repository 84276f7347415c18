"""SLAMTrainingArguments: the subset of transformers.TrainingArguments the reference's configs set
(config/training_args/default.yaml) + the two SLAM fields (slam_trainer.py:20-24)."""
import math
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class SLAMTrainingArguments:
    output_dir: str = "./results"
    learning_rate: float = 1e-3
    lr_scheduler_type: str = "cosine_with_min_lr"
    lr_scheduler_kwargs: dict = field(default_factory=lambda: {"min_lr": 5e-5})
    warmup_steps: int = 100
    warmup_ratio: float = 0.01
    max_grad_norm: float = 0.5
    num_train_epochs: float = 1.0
    max_steps: int = -1
    per_device_train_batch_size: int = 8
    per_device_eval_batch_size: int = 8
    gradient_accumulation_steps: int = 1
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    logging_steps: int = 10
    eval_strategy: str = "no"
    eval_steps: int = 1000
    save_steps: int = 0
    save_total_limit: int = 2
    seed: int = 42
    bf16: bool = True
    average_tokens_across_devices: bool = True   # transformers 5.x default (SURVEY.md §8c drift note)
    ddp_bucket_layers: int = 4                     # decoder layers per gradient all-reduce bucket
    ddp_comm_dtype: Optional[str] = "bfloat16"     # gradient buckets cross xGMI in bf16 (the reference's DDP precision: its gradients are bf16; half the bytes); "float32": reduce the fp32 buffer in place
    ddp_algo: str = "all_reduce"                   # "all_reduce": every rank reduces and updates everything (torch DDP's scheme); "rs_ag": reduce-scatter gradients, AdamW on the owned 1/N shard, all-gather bf16 parameters under the next forward (dp.ShardedGradReducer)
    optim_state_dtype: str = "float32"             # "float32": fp32 master weights + fp32 Adam moments (30 B/param per step); "bfloat16": the recipe's own precision (slam.yaml:9) - bf16 parameters and moments updated in place, no master (16 B/param); "float32_bf16_moments": fp32 master + bf16 moments (22 B/param)
    grad_dtype: Optional[str] = None               # precision the FINAL gradients of an optimizer step are kept in for the clip and AdamW: "bfloat16" = the reference's own (bf16 parameters have bf16 .grad, slam.yaml:9): the last backward stores them in bf16 only and emits the norm partials from the same stores; "float32"; None = bfloat16 with optim_state_dtype bfloat16, else float32. Micro-batches always accumulate in fp32
    grad_norm_from_backward: bool = True           # the clip's global norm from the sums of squares the last backward's final-value stores emit (no pass over the gradient buffer); False: the chunked norm pass - the summation order data-parallel runs use (they take the norm after the exchange), for bit-exact comparisons with them
    overwrite_first_grad: bool = True               # first backward of a step stores gradients (no zeroing pass); False = zero in AdamW
    overlap_optimizer: bool = False                # AdamW of the later layers under the next step's first layers (measured neutral: 272.7 vs 273.9 k tok/s)
    dataloader_num_workers: int = 0                # > 0: one background thread collates up to two optimizer steps ahead into pinned host memory (SLAMTrainer._micro_batches)
    min_token_id_count: Optional[int] = None
    max_token_id_count: Optional[int] = None
    # accepted for config compatibility, unused by the engine
    eval_accumulation_steps: Optional[int] = None
    use_cpu: bool = False
    ddp_find_unused_parameters: bool = False
    group_by_length: bool = False
    torch_compile: bool = False
    report_to: list = field(default_factory=list)
    run_name: Optional[str] = None

    def get_warmup_steps(self, num_training_steps: int) -> int:
        """TrainingArguments.get_warmup_steps: warmup_steps wins when > 0, else ceil(ratio * steps)."""
        return self.warmup_steps if self.warmup_steps > 0 else math.ceil(num_training_steps * self.warmup_ratio)


def lr_lambda(args: SLAMTrainingArguments, step: int, num_training_steps: int) -> float:
    """Multiplier of learning_rate at optimizer step `step` (0-based, before the update).
    cosine_with_min_lr: transformers/optimization.py:326-333 - linear warmup, then
    0.5(1+cos(pi p))(1-r)+r with r = min_lr/lr; `linear` and `constant` for completeness."""
    warm = args.get_warmup_steps(num_training_steps)
    if step < warm:
        return float(step) / float(max(1, warm))
    kind = args.lr_scheduler_type
    if kind == "constant" or kind == "constant_with_warmup":
        return 1.0
    prog = float(step - warm) / float(max(1, num_training_steps - warm))
    if kind == "linear":
        return max(0.0, 1.0 - prog)
    if kind == "cosine":
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
    if kind == "cosine_with_min_lr":
        kw = args.lr_scheduler_kwargs or {}
        if "min_lr" in kw and kw["min_lr"] is not None:
            r = kw["min_lr"] / args.learning_rate
        else:
            r = kw.get("min_lr_rate", 0.0)
        f = 0.5 * (1.0 + math.cos(math.pi * prog))
        return max(0.0, f * (1 - r) + r)
    raise ValueError(f"unsupported lr_scheduler_type {kind}")
