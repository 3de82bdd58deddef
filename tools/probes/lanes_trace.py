"""which operand of which hc_bwd call differs when the WGRAD lane is on?  eager schedule, the GPU held back by a spin kernel so
that the host queues the whole backward first (the launch cadence of a plan replay), checksums of every hc_bwd operand"""
import os, random, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT), str(ROOT / 'tests')]
import torch
from e2_tts_pytorch_amd import Transformer, ops
from test_backbone import randomize
dev = 'cuda'
random.seed(0); torch.manual_seed(0)
dim, depth, B, T = 512, 6, 4, 200
mod = Transformer(dim=dim, depth=depth, heads=dim // 64, dropout=0., max_seq_len=T)
randomize(mod); mod = mod.to(dev)
mod.enable_plans(False)
R = torch.randn(B, T, dim).to(dev)
trace = []
orig = {n: getattr(ops, n) for n in ('hc_bwd', 'rmsnorm_bwd', 'gemm_nt', 'dwconv_bwd', 'attn_bwd', 'qkv_post_bwd', 'geglu_bwd', 'gate_bwd')}
def cks(t):
    return None if t is None or not torch.is_tensor(t) else t.float().abs().sum()
def wrap(name):
    f = orig[name]
    def g(*a, **kw):
        ins = [cks(x) for x in a] + [cks(v) for v in kw.values()]
        out = f(*a, **kw)
        outs = [cks(x) for x in (out if isinstance(out, tuple) else (out,))]
        trace.append((name, ins, outs))
        return out
    return g
def run(lanes):
    mod.enable_lanes(lanes)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, dim, generator=g).to(dev).requires_grad_(True)
    t = torch.rand(B, generator=g).to(dev)
    txt = torch.randn(B, T, dim // 2, generator=g).to(dev).requires_grad_(True)
    mod.zero_grad(set_to_none=True)
    out = mod(x, times=t, text_embed=txt)
    loss = (out * R).sum()
    torch.cuda.synchronize()
    trace.clear()
    for n in orig: setattr(ops, n, wrap(n))
    torch.cuda._sleep(int(2e9))
    loss.backward()
    torch.cuda.synchronize()
    for n in orig: setattr(ops, n, orig[n])
    tr = [(n, [None if v is None else float(v) for v in i], [None if v is None else float(v) for v in o]) for n, i, o in trace]
    return x.grad.clone(), tr
dx0, t0 = run(False)
dx1, t1 = run(True)
print('dx rel', float((dx1 - dx0).norm() / dx0.norm()), 'calls', len(t0), len(t1))
shown = 0
for i, (a, b) in enumerate(zip(t0, t1)):
    if a != b:
        print(i, a[0], 'ins', [(x, y) for x, y in zip(a[1], b[1]) if x != y], 'outs', [(x, y) for x, y in zip(a[2], b[2]) if x != y])
        shown += 1
        if shown > 6: break
