"""TEST ONLY: point the package's ctypes binding at another build of the same C ABI -- the host logic-checker build of
the kernels (tests/emu/build_emu.py) or a stub -- and allow host pointers, or undo that.  The product module
(e2_tts_pytorch_amd/_lib.py) only holds the two variables; nothing in the package calls this."""


def install(path, host_pointers: bool):
    from e2_tts_pytorch_amd import _lib
    _lib._lib = _lib._Lib(path) if path is not None else None
    _lib._host_pointers_ok = bool(host_pointers)
