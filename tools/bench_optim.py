"""time the fused optimizer side (global-norm clip + ADOPT + EMA, K19) at cfg3 scale: ms and effective GB/s"""
import random, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
import bench
from e2_tts_pytorch_amd import E2TTS
from e2_tts_pytorch_amd.optim import FusedAdopt, FusedEMA, _runs
dim, depth, heads, B, T = bench.CONFIGS['cfg3']
random.seed(1234); torch.manual_seed(1234)
model = E2TTS(transformer=dict(dim=dim, depth=depth, heads=heads, dropout=0.1), use_vocos=False, cond_drop_prob=0.).cuda().train()
mel = torch.randn(2, 256, 100, device='cuda'); text = bench.synthetic_text(2, 1000)
opt = FusedAdopt(model, lr=1e-4, max_grad_norm=1.0)
ema = FusedEMA(model, update_after_step=0, update_every=1)
n = sum(p.numel() for p in opt.params)
def fb():
    out = model(mel, text=text); out.loss.backward()
for _ in range(2):
    fb(); opt.step(); opt.zero_grad(); ema.update()
fb()
pairs = [(p, p.grad) for p in opt.params if p.grad is not None]
print('parameters', n, 'tensors', len(pairs), 'generic runs', len(_runs(pairs)), '(the backbone is taken as one flat run)')
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record(); opt.step(); e1.record(); ema.update(); e2.record(); torch.cuda.synchronize()
t_opt, t_ema = e0.elapsed_time(e1), e1.elapsed_time(e2)
print(f'clip+adopt: {t_opt:.2f} ms = {n * 4 * 8 / t_opt / 1e6:.0f} GB/s (1 + 4 reads, 3 writes per element); ema: {t_ema:.2f} ms = {n * 4 * 3 / t_ema / 1e6:.0f} GB/s')
