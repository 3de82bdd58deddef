"""MI355X-native hot path of lucidrains/e2-tts-pytorch.

Same public names as the reference package (`e2_tts_pytorch/__init__.py:1-8`), minus the trainer harness.
"""
import sys as _sys
import types as _types

from .backbone import Transformer
from .e2_tts import E2TTS, DurationPredictor, MelSpec, E2TTSReturn, LossBreakdown

__all__ = ['Transformer', 'E2TTS', 'DurationPredictor', 'MelSpec', 'E2TTSReturn', 'LossBreakdown', 'install_as_reference', 'set_runtime_env']


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


def set_runtime_env():
    """The two HIP runtime settings the measured numbers were taken with; call BEFORE the process makes its first device call (the runtime
    reads them once).  GPU_MAX_HW_QUEUES=8: the launch lanes, the gradient-exchange stream and RCCL's streams each get a hardware queue
    (with the default 4 a fifth stream shares one and serialises: +11 ms per cfg3 step, profiles/r03_hw_queues.jsonl).
    HIP_FORCE_DEV_KERNARG=1: kernel arguments in device memory -- shorter gaps between the ~1600 dependent launches of a step (cfg3 -1.1 %,
    cfg2 -3.3 %, profiles/r06g_kernarg_ab.txt).  Values already present in the environment are left alone."""
    import os
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
