"""Ablation of the attention forward at the cfg3 shape (B 8, H 16, N 1056): the kernel with one stage switched off at a time
(E2K_ATTN_PROBE_* flags; results are wrong on purpose) -- which stage's removal buys how much.  -> gpurun_out/r02_attn_ablate.json"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
B, H, N = 8, 16, 1056
I = H * 64
torch.manual_seed(0)
qkvg = torch.randn(B * N, 3 * I + 2 * H, device=dev).to(bf16)
cosb, sinb = ops.rotary_table(N, dev)
st = ops.qkv_post_fwd(qkvg, B, H, N, cosb, sinb, None)
kmask, _ = ops.build_masks(None, B, N - 32, 32, dev, False)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
names = {0: 'full', 1: 'no kmask load', 2: 'no QK', 4: 'no softmax', 8: 'no PV', 16: 'no global loads', 32: 'no barriers',
         1 | 32: 'no kmask, no barriers', 1 | 16 | 32: 'no kmask/loads/barriers', 2 | 4 | 8: 'loads + barriers only', 1 | 4: 'no kmask, no softmax'}
rows = []
for p_drop in (0.0, 0.1):
    for rounds in range(2):
        for fl, nm in names.items():
            ops.attn_probe = fl
            ms = timeit(lambda: ops.attn_fwd(st, kmask, p_drop, 1, 3, None))
            if rounds == 1:
                rows.append(dict(p_drop=p_drop, probe=fl, name=nm, ms=round(ms, 4)))
                print(p_drop, nm, round(ms * 1e3, 1), 'us', flush=True)
ops.attn_probe = 0
json.dump(rows, open(ROOT / 'gpurun_out' / 'r02_attn_ablate.json', 'w'), indent=1)
