from .audio_tokeniser import AudioTokeniser, tokeniser_factory
from .unit_tokeniser import UnitTokeniser, WordLevelUnitVocab

__all__ = ["AudioTokeniser", "tokeniser_factory", "UnitTokeniser", "WordLevelUnitVocab"]
