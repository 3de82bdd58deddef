"""TEST INFRASTRUCTURE: compile the kernel sources against the host logic-checker runtime
(tests/emu/hip/hip_runtime.h) -> tests/emu/libe2k_emu.so.  Never used by the product path."""
from __future__ import annotations

import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / 'e2-tts-pytorch_amd' / 'csrc'
OUT = HERE / 'libe2k_emu.so'
OBJ = HERE / 'build'
CXX = os.environ.get('EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
FLAGS = ['-O2', '-g0', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-value', '-Wno-unknown-attributes', '-Wno-psabi',
         '-I', str(HERE), '-I', str(ROOT / 'include'), '-I', str(CSRC), '-x', 'c++']


def build(force: bool = False, verbose: bool = False) -> Path:
    srcs = sorted(CSRC.glob('*.hip'))
    hdrs = sorted(CSRC.glob('*.h')) + sorted(HERE.glob('*.h')) + [HERE / 'hip' / 'hip_runtime.h', ROOT / 'include' / 'e2k.h']
    OBJ.mkdir(exist_ok=True)
    hd = hashlib.sha256(b''.join(h.read_bytes() for h in hdrs) + ' '.join(FLAGS).encode()).hexdigest()

    def compile_one(src: Path):
        obj = OBJ / (src.stem + '.o')
        stamp = OBJ / (src.stem + '.stamp')
        d = hashlib.sha256(src.read_bytes() + hd.encode()).hexdigest()
        if not force and obj.exists() and stamp.exists() and stamp.read_text() == d:
            return obj, False
        cmd = [CXX, *FLAGS, '-c', str(src), '-o', str(obj)]
        if verbose:
            print('[emu build]', ' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        stamp.write_text(d)
        return obj, True

    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(compile_one, srcs))
    if any(ch for _, ch in res) or not OUT.exists() or force:
        subprocess.run([CXX, '-shared', '-fPIC', *[str(o) for o, _ in res], '-o', str(OUT), '-lpthread'], check=True)
    return OUT


if __name__ == '__main__':
    print(build(verbose=True))
