"""Which tensor-library ops (and how many hipMemcpy-class copies) does one REPLAYED cfg3 training step still issue outside the
recorded plan?  torch.profiler over one step after warm-up.  -> gpurun_out/step_torch_ops.json"""
import json, os, random, sys
from pathlib import Path
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synthetic_text
from e2_tts_pytorch_amd import E2TTS
dev = torch.device('cuda')
random.seed(1234); torch.manual_seed(1234)
dim, depth, heads, B, T = (int(a) for a in (sys.argv[1:6] or (1024, 24, 16, 8, 1024)))
model = E2TTS(transformer=dict(dim=dim, depth=depth, heads=heads, dropout=0.1), use_vocos=False, cond_drop_prob=0.).to(dev).train()
tr = model.transformer
tr.enable_persistent_grads()
flat = {id(q) for q, _ in tr._layout.slots}
params = [p for p in model.parameters() if id(p) not in flat]
mel = torch.randn(B, T, 100, device=dev)
text = synthetic_text(B, 1000)
def step():
    out = model(mel, text=text)
    out.loss.backward()
    for p in params:
        p.grad = None
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(({'name': e.key, 'count': e.count, 'cpu_ms': e.cpu_time_total / 1e3, 'device_ms': getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)) / 1e3} for e in ev),
              key=lambda r: -r['count'])
top = rows[:40]
for r in top:
    print(f"{r['count']:6d} cpu {r['cpu_ms']:8.2f} ms  dev {r['device_ms']:8.2f} ms  {r['name'][:90]}")
(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(rows[:120], open(ROOT / 'gpurun_out' / 'step_torch_ops.json', 'w'), indent=1)
