"""MI355X-native hot path of lucidrains/e2-tts-pytorch.

Same public names as the reference package (`e2_tts_pytorch/__init__.py:1-8`).
"""
from .backbone import Transformer
from .e2_tts import E2TTS, DurationPredictor, MelSpec, E2TTSReturn, LossBreakdown

__all__ = ['Transformer', 'E2TTS', 'DurationPredictor', 'MelSpec', 'E2TTSReturn', 'LossBreakdown']
