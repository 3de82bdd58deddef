"""depthwise-conv forward at the cfg3 audio / text shapes: us per call and the HBM rate of the algorithmic bytes.
(profiles/r03_dwconv_fwd_ab.jsonl was taken with this script on both forward kernels; the first one is gone.)"""
import json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
dev = torch.device('cuda')
torch.manual_seed(0)
res = {'variant': '128-frame tiles, weights + mask through LDS'}
for name, (B, N, C) in {'audio': (8, 1056, 1024), 'text': (8, 1056, 512)}.items():
    x = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    w = torch.randn(C, 31, device=dev) * 0.1
    bias = torch.randn(C, device=dev) * 0.1
    mask = torch.ones(B, N, dtype=torch.bool, device=dev); mask[1, 900:] = False
    for m, tag in ((None, 'nomask'), (mask, 'mask')):
        for _ in range(10):
            ops.dwconv_fwd(x, m, w, bias)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.dwconv_fwd(x, m, w, bias)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        res[f'{name}_{tag}_us'] = round(us, 2)
        res[f'{name}_{tag}_TBps'] = round(B * N * C * 6 / us / 1e6, 2)
print(json.dumps(res))
