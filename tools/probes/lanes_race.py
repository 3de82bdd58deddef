"""Diagnostic for tests/test_backbone.py::test_launch_lanes_match_single_stream: which tensor of which step differs between the single-stream
eager schedule and plans + lanes, and by how much (MI355X; E2K_* switches from the environment)."""
import os
import random
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / 'e2-tts-pytorch_amd'))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
from e2_tts_pytorch_amd import Transformer                     # noqa: E402
from test_backbone import randomize                            # noqa: E402

dev = 'cuda'
SEED = int(os.environ.get('SEED', '0'))      # (another model and other inputs: is the mismatch a property of particular values?)
random.seed(SEED)
torch.manual_seed(SEED)
dim, depth, B, T = 512, 6, 4, 200
mod = Transformer(dim=dim, depth=depth, heads=dim // 64, dropout=0., max_seq_len=T, num_registers=32)
randomize(mod, seed=SEED)
mod = mod.to(dev)
R = torch.randn(B, T, dim).to(dev)


def inputs(seed):
    g = torch.Generator().manual_seed(seed + 100 * SEED)
    x = torch.randn(B, T, dim, generator=g).to(dev).requires_grad_(True)
    t = torch.rand(B, generator=g).to(dev)
    txt = torch.randn(B, T, dim // 2, generator=g).to(dev).requires_grad_(True)
    return x, t, txt


def step(seed):
    mod.zero_grad(set_to_none=True)
    x, t, txt = inputs(seed)
    out = mod(x, times=t, text_embed=txt)
    (out * R).sum().backward()
    return dict(out=out.detach().clone(), dx=x.grad.clone(), dt=txt.grad.clone(), **{n: p.grad.clone() for n, p in mod.named_parameters()})


def infer(seed):
    with torch.no_grad():
        x, t, txt = inputs(seed)
        return dict(out=mod(x, times=t, text_embed=txt).clone())


def where(v, w):
    d = (v.float() - w.float()).abs()
    idx = d.reshape(d.shape[0], d.shape[1], -1).amax(-1).nonzero()
    return f'{int((d > 0).sum())} of {d.numel()} elements, max {float(d.max()):.3g} (|ref| max {float(w.float().abs().max()):.3g}); (batch, position) rows touched: {len(idx)}, first {idx[:6].tolist()}, last {idx[-3:].tolist()}'


def compare(tag, got, want):
    for n, v in got.items():
        w = want[n]
        if n in ('out', 'dx', 'dt'):
            if not torch.equal(v, w):
                print(f'{tag}: {n} differs: {where(v, w)}', flush=True)
        else:
            d = float((v.float() - w.float()).norm() / w.float().norm().clamp_min(1e-30))
            if d >= 2e-3 and float(w.float().norm()) >= 1e-6:
                print(f'{tag}: parameter gradient {n} rel-L2 {d:.3g}', flush=True)


seeds = (1, 2, 3, 2, 1)
mod.enable_lanes(False, backward=False)
mod.enable_plans(False)
ref_t = {s: step(s) for s in set(seeds)}
ref_i = {s: infer(s) for s in set(seeds)}
for rep in range(int(os.environ.get('REPS', '3'))):
    for lanes in (False, True):
        mod.enable_lanes(lanes, backward=lanes)
        for plans in (False, True):
            mod.enable_plans(plans)
            for i, s in enumerate(seeds):
                compare(f'rep {rep} lanes={lanes} plans={plans} training step {i} (seed {s})', step(s), ref_t[s])
            for i, s in enumerate(seeds):
                compare(f'rep {rep} lanes={lanes} plans={plans} no-grad forward {i} (seed {s})', infer(s), ref_i[s])
print('done', flush=True)
