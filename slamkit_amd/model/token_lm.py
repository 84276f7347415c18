"""TokenLM plugin surface - mirrors /root/reference slamkit/model/token_lm.py:7-43."""
from abc import ABC, abstractmethod
from typing import Optional

import torch


class TokenLM(ABC):
    @abstractmethod
    def log_likelihood(self, tokens: torch.Tensor, mean_nll: bool) -> torch.Tensor:
        """Per-sample log-likelihood of a right-padded token batch (token_lm.py:8-18)."""

    @abstractmethod
    def generate(self, inputs: Optional[torch.Tensor] = None, generation_config=None, **kwargs) -> torch.Tensor:
        """Continuation tokens for the given prompts (token_lm.py:20-27)."""


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def tlm_factory(cfg) -> TokenLM:
    """Same contract as the reference factory (token_lm.py:30-43): `cfg.tlm_type` in {twist, gslm},
    `cfg.pretrained_model` -> from_pretrained, else UnitLM(UnitLMConfig(**cfg.config_args))."""
    tlm_type = _get(cfg, "tlm_type")
    if tlm_type in ("twist", "gslm"):
        from .unit_lm import UnitLM, UnitLMConfig
        pretrained = _get(cfg, "pretrained_model")
        args = dict(_get(cfg, "config_args", {}) or {})
        if pretrained:
            return UnitLM.from_pretrained(pretrained)
        return UnitLM(UnitLMConfig(**args))
    raise ValueError(f"Unknown slm type: {tlm_type}")
