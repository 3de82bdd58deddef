"""cfg1 exactly as the reference's README runs it, on the HOST cores of the box (the CPU oracle: the reference itself cannot be
imported): E2TTS(dim=512, depth=8), mel = randn(2, 1024, 100), text = ['Hello', 'Goodbye'], one
forward + backward; median of 3 after one warm-up.  -> gpurun_out/r02_cpu_cfg1.json"""
import json, os, random, statistics, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from oracle import e2tts_oracle as O
threads = min(32, os.cpu_count() or 1)
torch.set_num_threads(threads)
random.seed(0); torch.manual_seed(0)
model = O.E2TTS(transformer=dict(dim=512, depth=8))
mel = torch.randn(2, 1024, 100)
text = ['Hello', 'Goodbye']
ts = []
for i in range(4):
    t0 = time.perf_counter()
    out = model(mel, text=text)
    out.loss.backward()
    ts.append(time.perf_counter() - t0)
    model.zero_grad()
med = statistics.median(ts[1:])
res = dict(config='cfg1 README: E2TTS(dim=512, depth=8), mel (2,1024,100), text [Hello, Goodbye], fwd+bwd, fp32 torch eager CPU oracle',
           threads=threads, host_cpus=os.cpu_count(), seconds_each=ts, seconds_median=med, mel_frames_per_s=2 * 1024 / med,
           step_tflop_algorithmic=1.115, tflops=1.115 / med)
print(json.dumps(res))
(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(res, open(ROOT / 'gpurun_out' / 'r02_cpu_cfg1.json', 'w'), indent=1)
