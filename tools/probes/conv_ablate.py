"""depthwise-conv backward at the cfg3 audio shape: workgroups per (channel tile, batch) sweep and ablations (no flush / loads + staging only).
profiles/r02_conv_ablate.json keeps the run that found the LDS float atomics (then in the flush) to be half the kernel: 92 us with them,
50 with plain LDS stores in their place, 38 without any flush."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops, _lib
bf16 = torch.bfloat16
dev = 'cuda'
B, N, C = 8, 1056, 1024
x = torch.randn(B, N, C, device=dev).to(bf16); w = torch.randn(C, 31, device=dev) * 0.1; bias = torch.zeros(C, device=dev)
pre, y = ops.dwconv_fwd(x, None, w, bias)
dy = torch.randn(B, N, C, device=dev).to(bf16)
dw = torch.zeros(C, 31, device=dev); db = torch.zeros(C, device=dev); dx = torch.empty_like(x)
L = _lib.get()
ws = torch.empty(17 * B * (C // 64) * 64 * 32, device=dev)
use_ws = True
def run(flag):
    L.e2k_dwconv_bwd(dy.data_ptr(), pre.data_ptr(), x.data_ptr(), None, w.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr() if use_ws else None, B, N, C, 31, flag, None)
def timeit(flag, iters=20):
    for _ in range(3): run(flag)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): run(flag)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
rows = []
for gx in (0, 2, 4, 9, 17):
    for tune, tag in ((0, 'full, workspace'), (-1, 'full, global atomics'), (1, 'no flush'), (3, 'loads + staging only, no flush')):
        use_ws = tune != -1
        tune = max(tune, 0)
        flag = ((tune | (gx << 7)) << 1)
        us = timeit(flag)
        rows.append(dict(gx=gx, variant=tag, us=round(us, 1)))
        print(rows[-1], flush=True)
json.dump(rows, open(ROOT / 'gpurun_out' / 'conv_ablate.json', 'w'), indent=1)
