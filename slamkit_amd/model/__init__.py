from .token_lm import TokenLM, tlm_factory
from .unit_lm import UnitLM, UnitLMConfig, CausalLMOutput

__all__ = ["TokenLM", "tlm_factory", "UnitLM", "UnitLMConfig", "CausalLMOutput"]
