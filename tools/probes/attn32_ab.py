"""Same-box A/B of the attention ring kernels at the bench shape (B 8, H 16, N 1056, dropout 0.1, keep masks handed over):
first generation (E2K_ATTN_RING16: 16 rows per wave) vs second generation (attn32.hip).  Interleaved rounds in one process; also checks the variants against each other
(outputs within bf16 rounding; shared masks == re-hashed masks bit for bit).  -> gpurun_out/r05_attn32_ab.json"""
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
dOg = torch.randn(M, I, device=dev).to(bf16)
VARIANTS = {'ring16': (64, '0'), 'attn32': (0, '0')}


def select(name):
    ops.attn_probe, os.environ['E2K_ATTN32_PROBE'] = VARIANTS[name]


def timeit(fn, iters=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


out = dict(shape=dict(B=B, H=H, N=N, p_drop=0.1), rounds=[], check={})
ref = {}
for name in VARIANTS:
    select(name)
    r = {}
    for share in (True, False):
        ops.attn_share_dropmask = share
        Og = ops.attn_fwd(st, kmask, 0.1, 7, 3).clone()
        dQ, dK, dV, dg = ops.attn_bwd(st, dOg, kmask, 0.1, 7, 3)
        r[share] = [t.clone() for t in (Og, dQ, dK, dV, dg)]
    ops.attn_share_dropmask = True
    torch.cuda.synchronize()
    out['check'][name + ': shared masks == re-hashed (bit-identical)'] = all(torch.equal(a, b) for a, b in zip(r[True], r[False]))
    ref[name] = r[True]
base = ref['ring16']
for name in VARIANTS:
    if name != 'ring16':
        out['check'][name + ': max |x - ring16| / max |ring16| (Og, dQ, dK, dV, dgate)'] = [
            float((a.float() - b.float()).abs().max() / b.float().abs().max()) for a, b in zip(ref[name], base)]
print(json.dumps(out['check'], indent=1), flush=True)
for rnd in range(4):
    rec = {}
    for name in VARIANTS:
        select(name)
        for pd, tag in ((0.1, 'drop'), (0.0, 'nodrop')):
            ops.attn_fwd(st, kmask, pd, 7, 3)
            rec[f'{name} fwd {tag} us'] = round(timeit(lambda: ops.attn_fwd(st, kmask, pd, 7, 3)), 1)
            rec[f'{name} bwd {tag} us'] = round(timeit(lambda: ops.attn_bwd(st, dOg, kmask, pd, 7, 3)), 1)
    out['rounds'].append(rec)
    print(rec, flush=True)
af = 4.0 * B * H * N * N * 64
med = {k: sorted(r[k] for r in out['rounds'])[len(out['rounds']) // 2] for k in out['rounds'][0]}
out['median_us'] = med
out['median_tflops'] = {k: round((af if ' fwd ' in k else 2.5 * af) / v / 1e6, 1) for k, v in med.items()}
print(json.dumps(out['median_us'], indent=1))
(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(out, open(ROOT / 'gpurun_out' / 'r05_attn32_ab.json', 'w'), indent=1)
