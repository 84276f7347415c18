"""DPO entry point: same stages as /root/reference cli/preference_alignment_train.py:18-65 on the HIP engine.

  python -m slamkit_amd.cli.preference_alignment_train data.train_path=prefs.jsonl \
         model.pretrained_model=/path/to/pretrained training_args.output_dir=/tmp/dpo
The frozen reference model is a second engine instance loaded from the same checkpoint.
"""
import logging
import os
import sys

import torch
import torch.distributed as dist

from ..data import init_preference_optimization_dataset
from ..model import tlm_factory
from ..tokeniser import tokeniser_factory
from ..trainer import DPOConfig, RunTimeStopperCallback, SLAMDPOTrainer
from ..utils.config import load_config, to_container

logger = logging.getLogger(__name__)


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(message)s")
    cfg = load_config("preference_alignment_train", list(argv if argv is not None else sys.argv[1:]))
    if cfg.tokeniser.tokeniser_type == "interleave":
        raise ValueError("Interleave tokeniser not supported for Preference Alignment yet")
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    tokeniser = tokeniser_factory(cfg.tokeniser)
    ds = init_preference_optimization_dataset(cfg.data)
    if cfg.model.config_args.vocab_size == -1:
        cfg.model.config_args.vocab_size = len(tokeniser.text_tokeniser)
    model = tlm_factory(cfg.model)
    ref_model = tlm_factory(cfg.model)
    if not cfg.model.get("pretrained_model"):
        ref_model.load_state_dict(model.state_dict(torch.float32))
    known = DPOConfig.__dataclass_fields__
    args = DPOConfig(**{k: v for k, v in to_container(cfg.training_args).items() if k in known})
    callbacks = [RunTimeStopperCallback(cfg.run_time)] if cfg.get("run_time") is not None else None
    trainer = SLAMDPOTrainer(model=model, ref_model=ref_model, args=args, train_dataset=ds["train"],
                             eval_dataset=ds.get("validation"), processing_class=tokeniser, callbacks=callbacks)
    state = trainer.train(resume_from_checkpoint=cfg.get("cont_training", None))
    if int(os.environ.get("RANK", 0)) == 0:
        model.save_pretrained(os.path.join(args.output_dir, "final"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return state


if __name__ == "__main__":
    main()
