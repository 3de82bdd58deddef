"""ctypes binding of the e2k C ABI (include/e2k.h).

The prototypes are parsed from the header itself, so the header stays the single source of truth.
There is NO fallback: if libe2k.so is missing, `get()` raises and every op in this package fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re
from pathlib import Path

_PKG = Path(__file__).resolve().parent
# E2K_LIB: another build of the SAME ABI (tools/ab/build_variants.py: same-box A/B of kernel versions); never a fallback
_LIB_PATH = Path(os.environ['E2K_LIB']) if os.environ.get('E2K_LIB') else _PKG / 'libe2k.so'


def _find_header() -> Path:
    env = os.environ.get('E2K_HEADER')
    if env:
        return Path(env)
    for up in (_PKG.parent.parent, _PKG.parent, _PKG):
        cand = up / 'include' / 'e2k.h'
        if cand.exists():
            return cand
    raise FileNotFoundError('include/e2k.h not found (set E2K_HEADER)')


_CTYPES = {
    'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'double': ctypes.c_double,
    'uint32_t': ctypes.c_uint32, 'uint64_t': ctypes.c_uint64, 'unsigned': ctypes.c_uint32,
}


_RET = {}          # entry point -> ctypes return type (int status / int or int64_t query result)


def parse_header(path: Path | None = None):
    """-> {name: [ctype, ...]} for every `int e2k_*(...)` prototype in the header."""
    text = Path(path or _find_header()).read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    protos = {}
    for m in re.finditer(r'\b(int|int64_t)\s+(e2k_\w+)\s*\(([^)]*)\)\s*;', text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        _RET[name] = _CTYPES[ret]
        types = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    types.append(ctypes.c_void_p)
                else:
                    toks = [t for t in a.replace('const', ' ').split() if t]
                    types.append(_CTYPES[toks[0]])
        protos[name] = types
    return protos


class E2KError(RuntimeError):
    pass


class _Lib:
    def __init__(self, path):
        self.path = str(path)
        self.cdll = ctypes.CDLL(self.path)
        self.protos = parse_header()
        for name, types in self.protos.items():
            fn = getattr(self.cdll, name)        # AttributeError here == header/library mismatch: fail loudly
            fn.argtypes = types
            fn.restype = _RET.get(name, ctypes.c_int)
            if name.startswith('e2k_query_') or name == 'e2k_version':
                setattr(self, name, fn)           # returns a value, not a status
            else:
                setattr(self, name, self._wrap(name, fn))

    @staticmethod
    def _wrap(name, fn):
        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise E2KError(f'{name} failed with code {rc}')
        call.__name__ = name
        return call


_lib = None
_host_pointers_ok = False      # only the test-side logic checker sets this (tests/emu/install.py)


def get() -> _Lib:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise E2KError(f'{_LIB_PATH} is missing: build it with `python __graft_entry__.py` '
                           '(hipcc --offload-arch=gfx950); there is no CPU / PyTorch fallback')
        _lib = _Lib(_LIB_PATH)
    return _lib


def host_pointers_ok() -> bool:
    return _host_pointers_ok
