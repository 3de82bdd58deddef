"""launch-lane fault isolation: steps of a mid-size backbone under (lanes, plans) settings, mismatches against the single-stream eager schedule"""
import random, sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT), str(ROOT / 'tests')]
import torch
from e2_tts_pytorch_amd import Transformer
from test_backbone import randomize
dev = 'cuda'
random.seed(0); torch.manual_seed(0)
dim, depth, B, T = (1024, 4, 8, 1024) if os.environ.get('SIZE') == 'big' else ((512, 6, 4, 224) if os.environ.get('SIZE') == 'm64' else (512, 6, 4, 200))
mod = Transformer(dim=dim, depth=depth, heads=dim // 64, dropout=0., max_seq_len=T)
randomize(mod); mod = mod.to(dev)
R = torch.randn(B, T, dim).to(dev)
def inputs(seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, T, dim, generator=g).to(dev).requires_grad_(True), torch.rand(B, generator=g).to(dev),
            torch.randn(B, T, dim // 2, generator=g).to(dev).requires_grad_(True))
def step(seed):
    mod.zero_grad(set_to_none=True)
    x, t, txt = inputs(seed)
    out = mod(x, times=t, text_embed=txt)
    (out * R).sum().backward()
    torch.cuda.synchronize()
    return out.detach().clone(), x.grad.clone(), txt.grad.clone(), {n: p.grad.clone() for n, p in mod.named_parameters()}
def rel(a, b): return float((a - b).norm() / b.norm().clamp_min(1e-30))
seeds = (1, 2, 3, 2, 1, 3)
res = {}
for lanes in (False, True):
    mod.enable_lanes(lanes)
    for plans in (False, True):
        mod.enable_plans(plans)
        res[lanes, plans] = [step(s) for s in seeds]
ref = res[False, False]
for key in ((False, True), (True, False), (True, True)):
    for i, ((o0, dx0, dt0, g0), (o1, dx1, dt1, g1)) in enumerate(zip(ref, res[key])):
        worst = max(((rel(g1[n], g0[n]), n) for n in g0 if float(g0[n].norm()) > 1e-6))
        print(key, 'step', i, 'out', rel(o1, o0), 'dx', rel(dx1, dx0), 'dt', rel(dt1, dt0), 'worst param', worst, flush=True)
