"""A/B of the weight-gradient kernels on the cfg3 shapes: 128 x 128 (mode 2) vs 256 x 256 8-phase (mode 3), interleaved,
HIP events, several token-split counts for the 256 kernel.  -> gpurun_out/r02_tn_ab.json"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
M1, M4 = 8448, 33792
shapes = [(M1, 8192, 1024, 'ff1'), (M1, 1024, 4096, 'ff2'), (M1, 3104, 1024, 'qkv'), (M1, 1024, 1024, 'attn out'),
          (M4, 1024, 1024, 'skip / cross a<-a'), (M4, 1024, 512, 'cross a<-t'), (M4, 512, 1024, 'cross t<-a'), (M4, 512, 512, 'cross t<-t'),
          (M1, 4096, 512, 'text ff1'), (M1, 512, 2048, 'text ff2'), (M1, 3104, 512, 'text qkv'), (M1, 512, 1024, 'text out')]
rows = []
def timeit(fn, iters=12):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, N, K, tag) in shapes:
    a = torch.randn(M, N, device=dev).to(bf16); b = torch.randn(M, K, device=dev).to(bf16)
    out = torch.zeros(N, K, device=dev)
    ref = None
    fl = 2.0 * M * N * K
    row = dict(tag=tag, M=M, N=N, K=K)
    ms = timeit(lambda: ops.gemm_tn(a, b, out, use_tr=2))
    row['k128'] = dict(ms=round(ms, 4), tf=round(fl / ms / 1e9, 1), splits=ops.lib().e2k_query_gemm_tn_splits_mode(M, N, K, 0, 2))
    out.zero_(); ops.gemm_tn(a, b, out, use_tr=2); ref = out.clone()
    best = None
    tiles = -(-N // 256) * -(-K // 256)
    cands = sorted({s for s in (0, max(1, 128 // tiles), max(1, 256 // tiles), max(1, 512 // tiles), max(1, 384 // tiles)) if s == 0 or s <= M // 256})
    for sp in cands:
        ms = timeit(lambda: ops.gemm_tn(a, b, out, use_tr=3, splits=sp))
        ns = ops.lib().e2k_query_gemm_tn_splits_mode(M, N, K, sp, 3)
        out.zero_(); ops.gemm_tn(a, b, out, use_tr=3, splits=sp)
        err = float((out - ref).abs().max() / ref.abs().max())
        r = dict(req=sp, splits=ns, ms=round(ms, 4), tf=round(fl / ms / 1e9, 1), err=err)
        row.setdefault('k256', []).append(r)
        if best is None or ms < best['ms']: best = r
    row['best256'] = best
    rows.append(row)
    print(tag, M, N, K, 'k128', row['k128'], 'best256', best, flush=True)
json.dump(rows, open(ROOT / 'gpurun_out' / 'r02_tn_ab.json', 'w'), indent=1)
