import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / 'e2-tts-pytorch_amd', ROOT, ROOT / 'tests'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long-running CPU test')


@pytest.fixture(scope='session')
def emu_lib():
    """Host logic-checker build of the kernels (tests/emu). TEST ONLY -- never a product path."""
    from emu.build_emu import build
    return build()


@pytest.fixture()
def emu(emu_lib):
    from e2_tts_pytorch_amd import _lib
    _lib._install_for_tests(emu_lib, host_pointers=True)
    yield _lib.get()
    _lib._install_for_tests(None, host_pointers=False)
