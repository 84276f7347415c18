from .callbacks import (MaxTokensStopperCallback, RunTimeStopperCallback, TrainerCallback, TrainerControl,
                        TrainerState)
from .training_args import SLAMTrainingArguments, lr_lambda
from .slam_trainer import SLAMTrainer

__all__ = ["SLAMTrainer", "SLAMTrainingArguments", "RunTimeStopperCallback", "MaxTokensStopperCallback",
           "TrainerCallback", "TrainerControl", "TrainerState", "lr_lambda"]
