"""fp32 oracle experiment: which bf16 rounding point produces the early-layer weight-gradient shortfall of the
transformer_variant case?  Rounds selectively: residual streams forward / stream gradients backward / branch inputs+outputs."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ('e2-tts-pytorch_amd', '', 'tests', 'tools'):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from oracle import e2tts_oracle as O
from oracle.golden_weights import fill_params

c = torch.load(os.path.join(ROOT, 'tests', 'golden', 'reference_pinned.pt'), weights_only=False)['transformer_variant']
random.seed(0)
ref = fill_params(O.Transformer(**c['kw'], cond_on_time=True), c['weight_seed'])

def mk(fw, bw):
    class R(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x): return x.bfloat16().float() if fw else x.clone()
        @staticmethod
        def backward(ctx, g): return g.bfloat16().float() if bw else g
    return R.apply

def run():
    x = c['x'].clone().requires_grad_(True); t = c['text'].clone().requires_grad_(True)
    for p in ref.parameters(): p.grad = None
    out = ref(x, times=c['times'], mask=c['mask'], text_embed=t)
    (out * c['R']).sum().backward()
    return {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}

g0 = run()
hc0 = O.HyperConnections.forward
names = ['layers.0.0.3.to_q.weight', 'layers.0.0.3.to_out.weight', 'layers.0.0.7.ff.2.weight', 'layers.1.0.3.to_q.weight', 'layers.1.1.2.to_out.weight']
for label, (sf, sb, bf, bb) in {'streams fwd only': (1, 0, 0, 0), 'stream grads only': (0, 1, 0, 0), 'branch in/out fwd': (0, 0, 1, 0),
                               'branch in/out bwd': (0, 0, 0, 1), 'all four': (1, 1, 1, 1)}.items():
    S, Bq = mk(sf, sb), mk(bf, bb)
    def hc(self, residuals, S=S, Bq=Bq):
        b, add = hc0(self, S(residuals))
        return Bq(b), (lambda y: S(add(Bq(y))))
    O.HyperConnections.forward = hc
    try:
        g1 = run()
    finally:
        O.HyperConnections.forward = hc0
    print('%-20s' % label, ' '.join('%s %.3f' % (n.split('layers.')[1][:12], float(g1[n].abs().sum() / g0[n].abs().sum())) for n in names))
