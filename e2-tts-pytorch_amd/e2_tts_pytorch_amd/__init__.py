"""MI355X-native hot path of lucidrains/e2-tts-pytorch.

Same public names as the reference package (`e2_tts_pytorch/__init__.py:1-8`), minus the trainer harness.
"""
import sys as _sys
import types as _types

from .backbone import Transformer
from .e2_tts import E2TTS, DurationPredictor, MelSpec, E2TTSReturn, LossBreakdown

__all__ = ['Transformer', 'E2TTS', 'DurationPredictor', 'MelSpec', 'E2TTSReturn', 'LossBreakdown', 'install_as_reference']


def install_as_reference():
    """Make `from e2_tts_pytorch.e2_tts import E2TTS, DurationPredictor, MelSpec` (trainer.py:29-33) and
    `from e2_tts_pytorch import E2TTS, ...` (train_example.py) resolve to this implementation.
    Call it before importing the reference's trainer; see INTEGRATION.md."""
    from . import e2_tts as _impl
    pkg = _types.ModuleType('e2_tts_pytorch')
    pkg.__path__ = []
    for name in ('Transformer', 'E2TTS', 'DurationPredictor', 'MelSpec'):
        setattr(pkg, name, globals()[name])
    pkg.e2_tts = _impl
    _sys.modules['e2_tts_pytorch'] = pkg
    _sys.modules['e2_tts_pytorch.e2_tts'] = _impl
    return pkg
