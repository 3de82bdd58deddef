"""What bounds attn_fwd32_kernel?  Bottleneck probes (E2K_ATTN32_PROBE bits: results are wrong on purpose) at the bench shape, dropout 0.1,
masks published by scalar stores.  1 no exp2, 2 no counter hash, 4 no score MFMAs, 8 no LDS fragment reads, 16 no output MFMAs,
32 no LDS-DMA after the first two tiles, 64 no barriers.  -> gpurun_out/r05_attn32_ablate.json"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch  # noqa: E402

from e2_tts_pytorch_amd import ops  # noqa: E402

bf16 = torch.bfloat16
dev = 'cuda'
B, H, N = 8, 16, 1056
M, I = B * N, H * 64
torch.manual_seed(0)
qkvg = torch.randn(M, 3 * I + 2 * H, device=dev).to(bf16)
cosb, sinb = ops.rotary_table(N, dev)
vfirst = torch.randn(B, H, N, 64, device=dev).to(bf16)
st = ops.qkv_post_fwd(qkvg, B, H, N, cosb, sinb, vfirst)
kmask = torch.zeros(B, st.Npad, dtype=torch.uint8, device=dev)
kmask[:, :N] = 1


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


NAMES = {0: 'full kernel', 1: 'no exp2', 2: 'no hash', 3: 'no exp2, no hash', 4: 'no score MFMAs', 8: 'no LDS fragment reads', 12: 'no score MFMAs, no LDS reads',
         16: 'no output MFMAs', 28: 'no MFMAs, no LDS reads', 32: 'no LDS-DMA', 64: 'no barriers', 96: 'no LDS-DMA, no barriers',
         31: 'no exp2 / hash / MFMAs / LDS reads', 127: 'everything off'}
out = {}
for rnd in range(3):
    for probe, name in NAMES.items():
        os.environ['E2K_ATTN32_PROBE'] = str(probe)
        us = timeit(lambda: ops.attn_fwd(st, kmask, 0.1, 7, 3))
        out.setdefault(name, []).append(round(us, 1))
os.environ['E2K_ATTN32_PROBE'] = '0'
for k, v in out.items():
    print(f'{k:45s} {v}')
(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(dict(shape=dict(B=B, H=H, N=N, p_drop=0.1), us=out), open(ROOT / 'gpurun_out' / 'r05_attn32_ablate.json', 'w'), indent=1)
