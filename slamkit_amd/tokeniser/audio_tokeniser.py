"""AudioTokeniser ABC + tokeniser_factory: /root/reference slamkit/tokeniser/audio_tokeniser.py:9-121."""
from abc import ABC, abstractmethod
from typing import Dict, List, Optional


def _get(cfg, key, default=None):
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


class AudioTokeniser(ABC):
    text_tokeniser = None

    def __init__(self):
        pass

    @abstractmethod
    def audio_represent(self, wav, lens=None) -> List[Dict]: ...

    @abstractmethod
    def stringify_representation(self, reps: List[Dict], mode: str = "test") -> List[str]: ...

    @abstractmethod
    def string_tokenise(self, audio_repr, return_tensors: Optional[str] = None) -> dict: ...

    @abstractmethod
    def tokenise(self, wav, lens=None) -> dict: ...

    @abstractmethod
    def build_prompt(self, wav, lens=None, output_modality: Optional[str] = None) -> dict: ...

    @abstractmethod
    def prepare_sample(self, sample: dict, **kw): ...

    @abstractmethod
    def decode_sample(self, tokens, output_modality: str = "SPEECH"): ...

    @abstractmethod
    def get_ignore_tokens(self, used_token_modality: str): ...


def tokeniser_factory(cfg) -> AudioTokeniser:
    """audio_tokeniser.py:107-121. `params.num_units` follows `feature_extractor.num_units`; with
    `params.load_fe` the reference's own HubertFeatureExtractor must be supplied by the caller
    (it stays the reference PyTorch-ROCm path), so only load_fe=False is constructed here."""
    params = dict(_get(cfg, "params", {}) or {})
    fe = _get(cfg, "feature_extractor", None)
    if fe is not None and _get(fe, "num_units") is not None:
        params["num_units"] = _get(fe, "num_units")
    if params.get("load_fe", False):
        raise ValueError("load_fe=True needs the reference slamkit.feature_extractor (HuBERT); pass load_fe=False "
                         "for the prepare_tokens/train stages (config/train.yaml:9-11)")
    ttype = _get(cfg, "tokeniser_type", "unit")
    if ttype == "unit":
        from .unit_tokeniser import UnitTokeniser
        return UnitTokeniser(None, **params)
    if ttype == "interleave":
        from .interleaving_tokeniser import InterleavingTokeniser
        params.pop("bos_eos_token_id", None)
        return InterleavingTokeniser(None, **params)
    raise ValueError(f"Unknown tokeniser type: {ttype}")
