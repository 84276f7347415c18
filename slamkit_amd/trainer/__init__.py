from .callbacks import (MaxTokensStopperCallback, RunTimeStopperCallback, TrainerCallback, TrainerControl,
                        TrainerState)
from .training_args import SLAMTrainingArguments, lr_lambda
from .slam_trainer import SLAMTrainer
from .slam_dpo_trainer import DPOConfig, SLAMDPOTrainer

__all__ = ["SLAMTrainer", "SLAMDPOTrainer", "DPOConfig", "SLAMTrainingArguments", "RunTimeStopperCallback", "MaxTokensStopperCallback",
           "TrainerCallback", "TrainerControl", "TrainerState", "lr_lambda"]
