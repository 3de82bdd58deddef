"""Runs the tests that are parametrised for the host model only (`emu` fixture) against the real library on a GPU.

Some tests were written when no GPU time was left and were therefore pinned to the host model; their bodies are
device-independent and take their device from the module-level `EMU_ONLY_DEV`.  This runner installs the real library,
points that switch at 'cuda' and calls them -- a one-off check for the first GPU call of a round, after which the tests
in question should simply be switched to the `dev` fixture.
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('e2-tts-pytorch_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))

import torch  # noqa: E402

from e2_tts_pytorch_amd import _lib  # noqa: E402


def main():
    assert torch.cuda.is_available()
    _lib._install_for_tests(None, host_pointers=False)
    _lib.get()
    import test_backbone
    import test_e2tts
    test_backbone.EMU_ONLY_DEV = test_e2tts.EMU_ONLY_DEV = 'cuda'
    failed = 0
    jobs = [('test_backbone.test_persistent_grads', lambda: test_backbone.test_persistent_grads(None))]
    for case in ('no_text', 'empty_string', 'short_lens', 'one_key_tile', 'text_longer_than_audio'):
        jobs.append((f'test_e2tts.test_edge_inputs[{case}]', lambda case=case: test_e2tts.test_edge_inputs(None, case)))
    jobs.append(('test_e2tts.test_training_dropout_shared_masks', lambda: test_e2tts.test_training_dropout_shared_masks(None)))
    for name, fn in jobs:
        try:
            fn()
            print('PASSED', name, flush=True)
        except Exception:       # noqa: BLE001
            failed += 1
            print('FAILED', name, flush=True)
            traceback.print_exc()
    print(f'{len(jobs) - failed} passed, {failed} failed')
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
