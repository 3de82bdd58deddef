"""Persistent 256 x 256 NT kernel (E2K_GEMM_PERSIST = 64) against the one-tile-per-workgroup kernel on the cfg3 shapes that
have at least one whole round of 256 x 256 tiles: bit comparison, 30-launch reproducibility screen, interleaved timing
rounds (back-to-back launches timed with HIP events), with the epilogue operands of the model's calls
(bias / per-batch gate / residual / dual-K).  -> gpurun_out/nt_persist_ab.json"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
# (M, N, K1, K2, residual, bias)
shapes = [(8448, 8192, 1024, 0, 0, 1), (8448, 4096, 1024, 0, 0, 0), (8448, 3104, 1024, 0, 0, 1), (33792, 1024, 1024, 512, 1, 0),
          (33792, 1024, 1024, 1024, 0, 0), (33792, 1024, 1024, 0, 1, 0), (33792, 512, 1024, 512, 1, 0), (8448, 4096, 512, 0, 0, 1),
          (8448, 2048, 512, 0, 0, 0), (8192, 8192, 1024, 0, 0, 0), (4096, 4096, 4096, 0, 0, 0)]
VARIANTS = (('default', 0), ('t256', 128), ('persist', 128 | 64))
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
rows = []
for (M, N, K1, K2, rs, bs) in shapes:
    torch.manual_seed(M + N + K1)
    a = torch.randn(M, K1, device=dev).to(bf16); a2 = torch.randn(M, K2, device=dev).to(bf16) if K2 else None
    b = (torch.randn(N, K1 + K2, device=dev) * 0.05).to(bf16)
    bias = torch.randn(N, device=dev) if bs else None
    resid = torch.randn(M, N, device=dev).to(bf16) if rs else None
    fl = 2.0 * M * N * (K1 + K2)
    row = dict(M=M, N=N, K1=K1, K2=K2, resid=rs, bias=bs)
    outs = {}
    for tag, f in VARIANTS:
        ops.gemm_flags = f
        outs[tag] = ops.gemm_nt(a, b, a2=a2, bias=bias, resid=resid).clone()
    row['persist_equals_t256_bits'] = bool(torch.equal(outs['persist'], outs['t256']))
    row['persist_vs_default_maxdiff'] = float((outs['persist'].float() - outs['default'].float()).abs().max())
    ops.gemm_flags = 128 | 64
    bad = 0
    for _ in range(30):
        bad += int(not torch.equal(ops.gemm_nt(a, b, a2=a2, bias=bias, resid=resid), outs['persist']))
    row['irreproducible_of_30'] = bad
    out = torch.empty(M, N, device=dev, dtype=bf16)
    t = {tag: [] for tag, _ in VARIANTS}
    for rnd in range(3):                                   # interleaved rounds, median reported
        for tag, f in VARIANTS:
            ops.gemm_flags = f
            t[tag].append(timeit(lambda: ops.gemm_nt(a, b, a2=a2, bias=bias, resid=resid, out=out)))
    for tag, _ in VARIANTS:
        ms = sorted(t[tag])[1]
        row[tag] = dict(us=round(ms * 1e3, 1), tf=round(fl / ms / 1e9, 1))
    ops.gemm_flags = 0
    rows.append(row)
    print(row, flush=True)
Path(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(rows, open(ROOT / 'gpurun_out' / 'nt_persist_ab.json', 'w'), indent=1)
