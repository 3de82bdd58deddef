"""Same-box A/B of kernel versions: boxes of the GPU pool differ by up to 7 % among themselves (the same tree ran a cfg3 step
in 85.5 and in 88.8 ms on two leases), so a kernel change can only be judged against its predecessor INSIDE one gpurun call.
This builds variants of libe2k.so in which ONE source file is taken from another commit (or a patched copy) while every other
object is today's: `name=path/to/file.hip@commit` (git show) or `name=path/to/other_copy.hip:as_file.hip`.  The libraries go to
gpurun_out-independent, git-ignored `tools/ab/lib/libe2k_<name>.so` (they travel to the GPU box with the snapshot);
E2K_LIB=<path> makes e2_tts_pytorch_amd/_lib.py load one (benchmark instrument: the default library is untouched).

    python tools/ab/build_variants.py r03=e2-tts-pytorch_amd/csrc/attn.hip@f7fd506 hash=e2-tts-pytorch_amd/csrc/attn.hip@e030591
"""
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
PKG = ROOT / 'e2-tts-pytorch_amd'
sys.path.insert(0, str(PKG))
import build_kernels as bk          # noqa: E402


def main():
    bk.build(verbose=False)         # today's objects
    out = ROOT / 'tools' / 'ab' / 'lib'
    out.mkdir(parents=True, exist_ok=True)
    for spec in sys.argv[1:]:
        name, what = spec.split('=', 1)
        if '@' in what:
            path, commit = what.split('@')
            text = subprocess.run(['git', 'show', f'{commit}:{path}'], cwd=ROOT, check=True, capture_output=True, text=True).stdout
            fname = Path(path).name
        else:
            src, fname = what.split(':')
            text = Path(src).read_text()
        with tempfile.TemporaryDirectory() as td:
            f = Path(td) / fname
            f.write_text(text)
            obj = Path(td) / (f.stem + '.o')
            subprocess.run([bk.HIPCC, *bk.FLAGS, '-c', str(f), '-o', str(obj)], check=True)
            objs = [obj if o.stem == f.stem else o for o in sorted(bk.OBJ.glob('*.o'))]
            lib = out / f'libe2k_{name}.so'
            subprocess.run([bk.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', *map(str, objs), '-o', str(lib)], check=True)
            print(lib)


if __name__ == '__main__':
    main()
