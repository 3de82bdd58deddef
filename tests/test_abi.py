"""The C-ABI library loads without a GPU and exports every symbol include/e2k.h declares (no compute calls here)."""
import ctypes
from pathlib import Path

import pytest


def install_lib(path, host_pointers):
    from emu.install import install
    install(path, host_pointers)


ROOT = Path(__file__).resolve().parent.parent


def test_header_parses_and_library_exports_everything():
    from e2_tts_pytorch_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 20 and 'e2k_gemm_nt_bf16' in protos and 'e2k_attn_bwd' in protos
    so = ROOT / 'e2-tts-pytorch_amd' / 'e2_tts_pytorch_amd' / 'libe2k.so'
    if not so.exists():
        import importlib.util
        spec = importlib.util.spec_from_file_location('build_kernels', ROOT / 'e2-tts-pytorch_amd' / 'build_kernels.py')
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    lib = ctypes.CDLL(str(so))
    missing = [n for n in protos if not hasattr(lib, n)]
    assert not missing, missing
    lib.e2k_version.restype = ctypes.c_int
    assert lib.e2k_version() >= 1


def test_no_cpu_fallback():
    """product ops refuse CPU tensors (the host logic-checker is only reachable through the test fixtures)"""
    import torch
    from e2_tts_pytorch_amd import _lib, ops
    install_lib(None, host_pointers=False)
    with pytest.raises(_lib.E2KError):
        ops.gemm_nt(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))


def test_install_as_reference_aliases():
    import sys
    import e2_tts_pytorch_amd as pkg
    saved = {k: sys.modules.get(k) for k in ('e2_tts_pytorch', 'e2_tts_pytorch.e2_tts')}
    try:
        pkg.install_as_reference()
        from e2_tts_pytorch.e2_tts import E2TTS, DurationPredictor, MelSpec   # the import line of trainer.py:29-33
        from e2_tts_pytorch import Transformer
        assert E2TTS is pkg.E2TTS and DurationPredictor is pkg.DurationPredictor and MelSpec is pkg.MelSpec
        assert Transformer is pkg.Transformer
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_module_contract():
    """what trainer.py touches: deepcopy (EMA), state_dict keys, parameters()"""
    import copy
    import random
    import torch
    import e2_tts_pytorch_amd as pkg
    from oracle import e2tts_oracle as O
    random.seed(0)
    torch.manual_seed(0)
    kw = dict(dim=256, depth=2, heads=4)
    m = pkg.E2TTS(transformer=dict(**kw), use_vocos=False)
    ref = O.E2TTS(transformer=dict(**kw))
    assert set(m.state_dict().keys()) == set(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert m.state_dict()[k].shape == v.shape, k
    m2 = copy.deepcopy(m)
    assert sum(p.numel() for p in m2.parameters()) == sum(p.numel() for p in ref.parameters())
    assert m.velocity_consistency_weight == 0.
    # the constructor's default-off switches are built and give the reference's state_dict keys; what stays refused says so
    ref_v = O.Transformer(dim=256, depth=2, has_freq_axis=True, attn_laser=True, attn_fourier_embed_input=True)
    mod_v = pkg.Transformer(dim=256, depth=2, has_freq_axis=True, attn_laser=True, attn_fourier_embed_input=True)
    assert {k: tuple(v.shape) for k, v in mod_v.state_dict().items()} == {k: tuple(v.shape) for k, v in ref_v.state_dict().items()}
    with pytest.raises(NotImplementedError):
        pkg.Transformer(dim=256, depth=2, dim_head=32)
    with pytest.raises(NotImplementedError):
        pkg.Transformer(dim=256, depth=2, num_residual_streams=2)
    with pytest.raises(AssertionError):
        pkg.Transformer(dim=256, depth=3)
