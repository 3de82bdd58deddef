"""host-model experiment: per-parameter sum|g| of the HIP path vs the reference golden for the transformer_variant case"""
import random, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT), str(ROOT / 'tests')]
import torch
from e2_tts_pytorch_amd import Transformer, _lib
from oracle.golden_weights import fill_params
from emu.build_emu import build


def install_lib(path, host_pointers):
    from emu.install import install
    install(path, host_pointers)


install_lib(build(), host_pointers=True)
case = sys.argv[1] if len(sys.argv) > 1 else 'transformer_variant'
c = torch.load(ROOT / 'tests' / 'golden' / 'reference_pinned.pt', weights_only=False)[case]
random.seed(0)
mod = fill_params(Transformer(**c['kw'], cond_on_time=c['cond_on_time']), c['weight_seed'])
mod.enable_plans(False)
x = c['x'].clone().requires_grad_(True)
t = c['text'].clone().requires_grad_(True) if c['text'] is not None else None
out = mod(x, times=c['times'], mask=c['mask'], text_embed=t)
(out * c['R']).sum().backward()
r2 = lambda a, b: float((a - b).norm() / b.norm())
print('out', r2(out.detach(), c['out']), 'dx', r2(x.grad, c['dx']))
rows = []
for n, p in mod.named_parameters():
    want = c['grad_abs_sums'].get(n)
    if want is None or p.grad is None or p.numel() < 16384 or want == 0.:
        continue
    rows.append((n, float(p.grad.double().abs().sum()) / want))
for n, r in rows:
    print(f'{r:6.3f}  {n}')
