import os
import sys
from pathlib import Path

import pytest


def install_lib(path, host_pointers):
    from emu.install import install
    install(path, host_pointers)


# the launch lanes + the gradient-exchange stream + RCCL's own streams need more than the default 4 hardware queues (DESIGN.md
# section 4.1 / 6); bench.py sets the same before its first device call, so tests and bench run one queue mapping
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')          # (kernel arguments in device memory: bench.py says why; the tests run the bench's runtime settings)

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / 'e2-tts-pytorch_amd', ROOT, ROOT / 'tests'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def gpu_shapes(dev):
    """the shapes a test uses on the MI355X; E2K_EMU_GPU_SHAPES=1 makes the (~1000x slower) host model run them too --
    used with tests/emu/guardmalloc.c to look for out-of-bounds accesses at exactly the sizes the GPU run sees"""
    return dev == 'cuda' or os.environ.get('E2K_EMU_GPU_SHAPES') == '1'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'late: run after everything else (the long full-size and 200-trial cases)')


def pytest_collection_modifyitems(config, items):
    """the comparisons with the REFERENCE'S OWN outputs (tests/golden/reference_pinned.pt) run first -- with -x a surprise
    anywhere else cannot hide them --, the long cases last"""
    def rank(it):
        if it.name.startswith('test_reference_golden'):
            return 0
        return 2 if it.get_closest_marker('late') else 1
    items.sort(key=rank)            # (stable: collection order inside each group)


@pytest.fixture(scope='session')
def emu_lib():
    """Host logic-checker build of the kernels (tests/emu). TEST ONLY -- never a product path."""
    from emu.build_emu import build
    return build()


@pytest.fixture()
def emu(emu_lib):
    from e2_tts_pytorch_amd import _lib
    install_lib(emu_lib, host_pointers=True)
    yield _lib.get()
    install_lib(None, host_pointers=False)


@pytest.fixture(params=['emu', pytest.param('gpu', marks=pytest.mark.gpu)])
def dev(request):
    """'cpu' with the host logic-checker build of the kernels, or 'cuda' with the real libe2k.so (-m gpu).
    The same test body checks the same kernels against the same oracle in both cases."""
    from e2_tts_pytorch_amd import _lib
    if request.param == 'emu':
        lib = request.getfixturevalue('emu_lib')
        install_lib(lib, host_pointers=True)
        yield 'cpu'
        install_lib(None, host_pointers=False)
    else:
        import torch
        assert torch.cuda.is_available(), 'gpu tests need a HIP device'
        install_lib(None, host_pointers=False)
        _lib.get()                      # raises if libe2k.so is missing: no silent fallback
        yield 'cuda'
