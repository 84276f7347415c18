"""Stopper callbacks - same behaviour as /root/reference slamkit/trainer/callbacks.py:10-54,
driven by the engine trainer's step loop instead of transformers.Trainer."""
import logging
import time
from dataclasses import dataclass
from typing import Union

logger = logging.getLogger(__name__)


@dataclass
class TrainerState:
    global_step: int = 0
    epoch: float = 0.0
    max_steps: int = 0
    num_input_tokens_seen: int = 0
    log_history: list = None

    def __post_init__(self):
        if self.log_history is None:
            self.log_history = []


@dataclass
class TrainerControl:
    should_training_stop: bool = False
    should_evaluate: bool = False
    should_save: bool = False
    should_log: bool = False


class TrainerCallback:
    def on_train_begin(self, args, state, control, **kwargs):
        pass

    def on_step_end(self, args, state, control, **kwargs):
        pass

    def on_train_end(self, args, state, control, **kwargs):
        pass


def parse_run_time(run_time: Union[str, int]) -> int:
    """'D-HH:MM:SS' or 'HH:MM:SS' or seconds (callbacks.py:14-26)."""
    if isinstance(run_time, int):
        return run_time
    days = 0
    if "-" in run_time:
        days, run_time = run_time.split("-")
        days = int(days)
    hours, minutes, seconds = run_time.split(":")
    return days * 24 * 60 * 60 + int(hours) * 60 * 60 + int(minutes) * 60 + int(seconds)


class RunTimeStopperCallback(TrainerCallback):
    """Stops (and asks for eval + save) once the wall-clock budget is spent (callbacks.py:10-39)."""

    def __init__(self, run_time: Union[str, int]):
        self.run_time = parse_run_time(run_time)

    def on_train_begin(self, args, state, control, **kwargs):
        logger.info(f"Training will run for {self.run_time} seconds")
        self.start_time = time.time()

    def on_step_end(self, args, state, control, **kwargs):
        if time.time() - self.start_time > self.run_time:
            control.should_training_stop = True
            control.should_evaluate = True
            control.should_save = True
            logger.info(f"Stopping training as it has run for {self.run_time} seconds")


class MaxTokensStopperCallback(TrainerCallback):
    """Stops once state.num_input_tokens_seen reaches the budget (callbacks.py:42-54)."""

    def __init__(self, train_max_tokens: int):
        self.max_tokens = train_max_tokens

    def on_train_begin(self, args, state, control, **kwargs):
        logger.info(f"Training will run for {self.max_tokens} tokens according to specified range if provided")

    def on_step_end(self, args, state, control, **kwargs):
        if state.num_input_tokens_seen >= self.max_tokens:
            control.should_training_stop = True
            control.should_evaluate = True
            control.should_save = True
            logger.info(f"Stopping training as it has seen {state.num_input_tokens_seen} tokens")
