"""list host<->device synchronisation points of a plan-mode training step (torch.cuda.set_sync_debug_mode)"""
import random, sys, time, warnings
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
import bench
from e2_tts_pytorch_amd import E2TTS
dim, depth, heads, B, T = bench.CONFIGS['cfg3']
random.seed(1234); torch.manual_seed(1234)
model = E2TTS(transformer=dict(dim=dim, depth=depth, heads=heads, dropout=0.1), use_vocos=False, cond_drop_prob=0.).cuda().train()
mel = torch.randn(B, T, 100, device='cuda'); text = bench.synthetic_text(B, 1000)
def step():
    out = model(mel, text=text); out.loss.backward(); model.zero_grad(set_to_none=True); return out.loss
for _ in range(3): step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    step()
torch.cuda.set_sync_debug_mode('default')
for x in w: print('SYNC:', str(x.message)[:100], '@', x.filename.split('/')[-1], x.lineno)
torch.cuda.synchronize()
# host time of the pieces
t0 = time.perf_counter(); out = model(mel, text=text); t1 = time.perf_counter(); out.loss.backward(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f'host: forward {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, drain {1e3*(t3-t2):.1f} ms')
