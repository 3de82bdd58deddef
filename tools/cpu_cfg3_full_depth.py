"""CPU baseline at the FULL depth of the headline config, once: the oracle (fp32 torch eager on the box's host cores) at
dim 1024 / depth 24 / 16 heads, T = 1024, B = 1, one forward + backward (B = 8 materialises eight 1056 x 1056 x 16 score
tensors per attention and all saved activations: it does not fit the bounded time of the bench's cpu_baseline leg, which
times 12 of the 24 layers and scales).  -> gpurun_out/r03_cpu_cfg3_full_depth.json"""
import json, os, random, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from oracle import e2tts_oracle as O
threads = min(32, os.cpu_count() or 1)
torch.set_num_threads(threads)
random.seed(0); torch.manual_seed(0)
ref = O.E2TTS(transformer=dict(dim=1024, depth=24, heads=16, dropout=0.1), cond_drop_prob=0.)
mel = torch.randn(1, 1024, 100)
t0 = time.perf_counter()
out = ref(mel, text=['The quick brown fox jumps over the lazy dog.'])
out.loss.backward()
dt = time.perf_counter() - t0
res = dict(config='dim 1024, depth 24, 16 heads, T 1024, B 1, fp32 torch eager (oracle), one forward + backward, first call (no warm-up)',
           threads=threads, host_cores=os.cpu_count(), seconds=dt, mel_frames_per_s=1024 / dt, loss=float(out.loss))
print(json.dumps(res))
(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(res, open(ROOT / 'gpurun_out' / 'r03_cpu_cfg3_full_depth.json', 'w'), indent=1)
