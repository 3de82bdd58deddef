"""Compile csrc/*.hip into e2_tts_pytorch_amd/libe2k.so for gfx950 (hipcc cross-compiles without a GPU).

Usage: python build_kernels.py [--force]
The .so is built in-tree so that it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
CSRC = HERE / 'csrc'
OUT = HERE / 'e2_tts_pytorch_amd' / 'libe2k.so'
OBJ = HERE / 'build'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast',
         '-I', str(ROOT / 'include'), '-I', str(CSRC)]


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    srcs = sorted(CSRC.glob('*.hip'))
    deps = srcs + sorted(CSRC.glob('*.h')) + [ROOT / 'include' / 'e2k.h']
    OBJ.mkdir(exist_ok=True)
    stamp = OBJ / 'stamp'
    dig = _digest(deps)
    if not force and OUT.exists() and stamp.exists() and stamp.read_text() == dig:
        return OUT
    hdr_dig = _digest([d for d in deps if d.suffix == '.h'])

    def compile_one(src: Path):
        obj = OBJ / (src.stem + '.o')
        ostamp = OBJ / (src.stem + '.stamp')
        d = hashlib.sha256(src.read_bytes() + hdr_dig.encode()).hexdigest()
        if not force and obj.exists() and ostamp.exists() and ostamp.read_text() == d:
            return obj
        cmd = [HIPCC, *FLAGS, '-c', str(src), '-o', str(obj)]
        if verbose:
            print('[e2k build]', ' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        ostamp.write_text(d)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', *map(str, objs), '-o', str(OUT)]
    if verbose:
        print('[e2k build]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(dig)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
