"""cfg5: E2TTS.sample() ODE inference (32 midpoint steps = 62 function evaluations x (cond + null) = 124 backbone
forwards), B=32, prompt 5 frames, duration 1024, cfg_strength 1, on one MI355X.  Reports sampled mel-frames/s."""
import json
import os
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch  # noqa: E402

from bench import synthetic_text  # noqa: E402
from e2_tts_pytorch_amd import E2TTS  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
B = int(args[0]) if len(args) > 0 else 32
steps = int(args[1]) if len(args) > 1 else 32
dev = 'cuda'
random.seed(0)
torch.manual_seed(0)
model = E2TTS(transformer=dict(dim=1024, depth=24, heads=16), use_vocos=False).to(dev)
if '--eager' in sys.argv:
    model.transformer.enable_plans(False)  # default: forward-only launch plans (one per input signature: cond pass / null pass)
cond = torch.randn(B, 5, 100, device=dev)
text = synthetic_text(B, 3)
model.sample(cond, text=text, duration=1024, steps=3)          # warm-up (packs weights, builds shadows, records the launch plans)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = model.sample(cond, text=text, duration=1024, steps=steps, cfg_strength=1.)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fwd_flops = 6336e12 * (B / 32) * ((steps - 1) / 31)
res = {'metric': 'sampled mel-frames/sec (E2TTS.sample, 32 midpoint steps, CFG)', 'value': B * 1024 / dt,
       'unit': 'mel-frames/s', 'seconds': dt, 'B': B, 'steps': steps, 'finite': bool(torch.isfinite(out).all()),
       'launch_plans': '--eager' not in sys.argv, 'cfg_passes_concurrent': __import__('e2_tts_pytorch_amd.e2_tts', fromlist=['x'])._CFG_CONCURRENT,
       'model_tflops_per_s': fwd_flops / dt / 1e12, 'shape': list(out.shape)}
print(json.dumps(res))
(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(res, open(ROOT / 'gpurun_out' / ('r06_sample_cfg5%s%s.json' % ('_eager' if '--eager' in sys.argv else '', '_sequential' if os.environ.get('E2K_CFG_CONCURRENT') == '0' else '')), 'w'), indent=1)
