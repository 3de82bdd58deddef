"""MelSpec kernel throughput at the benchmark shape: B = 8 utterances of 261 888 samples (1024 frames each), against its
algorithmic bytes 4 * nw + 400 * frames per utterance.  -> gpurun_out/r03_melspec.json"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import MelSpec
dev = 'cuda'
res = []
for B in (8, 64):
    nw = 261_888
    wave = torch.randn(B, nw, device=dev)
    m = MelSpec().to(dev)
    for _ in range(3): out = m(wave)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    iters = 20
    for _ in range(iters): out = m(wave)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    frames = out.shape[-1]
    nbytes = B * (4 * nw + 400 * frames)
    res.append(dict(B=B, samples=nw, frames=frames, ms=ms, algorithmic_MB=nbytes / 1e6, GBps=nbytes / ms / 1e6, frac_of_8TBps=nbytes / ms / 1e6 / 8000,
                    mel_frames_per_s=B * frames / ms * 1e3))
    print(res[-1], flush=True)
json.dump(res, open(ROOT / 'gpurun_out' / 'r03_melspec.json', 'w'), indent=1)
