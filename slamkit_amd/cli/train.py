"""Pre-training entry point: same stages and rules as /root/reference cli/train.py:16-89
(tokeniser -> dataset -> model -> SLAMTrainer.train) on the HIP engine.

  python -m slamkit_amd.cli.train data.train_path=example_data/tokens.jsonl model=default \
         training_args.per_device_train_batch_size=2 training_args.output_dir=/tmp/run
Under torchrun (one process per GPU) the RCCL process group is created from RANK/WORLD_SIZE.
"""
import logging
import math
import os
import sys

import torch
import torch.distributed as dist

from ..data import init_dataset
from ..model import tlm_factory
from ..tokeniser import tokeniser_factory
from ..trainer import MaxTokensStopperCallback, RunTimeStopperCallback, SLAMTrainer, SLAMTrainingArguments
from ..utils.config import load_config, to_container

logger = logging.getLogger(__name__)


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(message)s")
    cfg = load_config("train", list(argv if argv is not None else sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    # train.py:18-22: interleaved data must be tokenised with the text tokeniser of the model it feeds
    if cfg.tokeniser.get("tokeniser_type") == "interleave":
        want = cfg.model.config_args.get("base_model_name")
        have = cfg.tokeniser.params.get("text_tokeniser_path")
        if have != want:
            logger.warning(f"Text tokeniser {have}, doesn't match model changing it to: {want}")
            cfg.tokeniser.params["text_tokeniser_path"] = want
    # train.py:25-28
    if cfg.get("train_max_tokens") is not None and (cfg.get("ds_token_size") or 0) > 0:
        cfg.training_args.num_train_epochs = (cfg.train_max_tokens / cfg.ds_token_size) * 1.01
    tokeniser = tokeniser_factory(cfg.tokeniser)
    ds, collator = init_dataset(cfg, tokeniser)
    if cfg.model.config_args.vocab_size == -1:  # train.py:39-41
        cfg.model.config_args.vocab_size = len(tokeniser.text_tokeniser)
    bs = cfg.training_args.per_device_train_batch_size
    cfg.model.config_args["max_tokens"] = bs * cfg.model.context_len
    model = tlm_factory(cfg.model)
    if cfg.data.packing and model.config._attn_implementation != "flash_attention_2":  # train.py:43-45
        raise ValueError("Packing is only supported with flash_attention_2 model")
    # train.py:48-54: both warmup settings -> the larger one wins
    ta = cfg.training_args
    if (ta.get("warmup_steps", 0) or 0) > 0 and (ta.get("warmup_ratio", 0.0) or 0.0) > 0:
        gbs = bs * ta.gradient_accumulation_steps * world
        n_steps = math.ceil(len(ds["train"]) / gbs) * ta.num_train_epochs
        if n_steps * ta.warmup_ratio > ta.warmup_steps:
            ta.warmup_steps = 0
    known = SLAMTrainingArguments.__dataclass_fields__
    args = SLAMTrainingArguments(**{k: v for k, v in to_container(ta).items() if k in known})
    callbacks = []
    if cfg.get("run_time") is not None:
        callbacks.append(RunTimeStopperCallback(cfg.run_time))
    if cfg.get("train_max_tokens") is not None:
        callbacks.append(MaxTokensStopperCallback(cfg.train_max_tokens))
    trainer = SLAMTrainer(model=model, args=args, data_collator=collator, train_dataset=ds["train"],
                          eval_dataset=ds.get("validation"), processing_class=tokeniser, callbacks=callbacks)
    state = trainer.train(resume_from_checkpoint=cfg.cont_training)
    if int(os.environ.get("RANK", 0)) == 0:
        model.save_pretrained(os.path.join(args.output_dir, "final"))
        tokeniser.save_pretrained(os.path.join(args.output_dir, "final"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return state


if __name__ == "__main__":
    main()
