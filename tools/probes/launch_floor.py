"""How long does the HOST need for a training step when the GPU is not the bottleneck?  Same model depth (= same number of
launches per step) as cfg3, but a tiny batch / sequence, so that the kernels take almost no time: the step time is then the
launch floor of the scheduler.  Plan replay (C++) vs eager launches from Python."""
import json, random, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import E2TTS
dev = torch.device('cuda')
random.seed(0); torch.manual_seed(0)
dim, depth, heads = (int(a) for a in (sys.argv[1:4] or (256, 24, 4)))
B, T = 1, 96
model = E2TTS(transformer=dict(dim=dim, depth=depth, heads=heads, dropout=0.1), use_vocos=False, cond_drop_prob=0.).to(dev).train()
tr = model.transformer
tr.enable_persistent_grads()
flat = {id(q) for q, _ in tr._layout.slots}
params = [p for p in model.parameters() if id(p) not in flat]
mel = torch.randn(B, T, 100, device=dev)
text = ['hello world']
def step():
    out = model(mel, text=text)
    out.loss.backward()
    for p in params:
        p.grad = None
res = {}
for mode in ('plan', 'eager'):
    tr.enable_plans(mode == 'plan')
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    res[mode] = dict(host_ms_per_step=t_host / n * 1e3, wall_ms_per_step=t_all / n * 1e3)
    if mode == 'plan':
        rows = tr.plan_profile()
        res[mode]['recorded_calls'] = len(rows)
        res[mode]['gpu_ms_sum_of_calls'] = sum(r['ms'] for r in rows)
print(json.dumps(dict(dim=dim, depth=depth, B=B, T=T, **res)))
