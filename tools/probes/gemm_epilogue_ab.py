"""NT GEMM: C tile through LDS in whole-line row segments (default) vs straight from the accumulator registers
(E2K_GEMM_NO_STAGE = 64) on the cfg3 shapes of both NT kernels; weight-gradient GEMM timings (partials in fragment order
+ tn_reduce_frag_kernel) on the cfg3 shapes, to be read next to profiles/r03_tn_self_reduce_LOST.json's `two_launches`
column (the row-major partial layout, same boxes' class).  -> gpurun_out/gemm_epilogue_ab.json"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
# (M, N, K1, K2, residual, bias)
nt_shapes = [(8448, 8192, 1024, 0, 0, 1), (8448, 1024, 8192, 0, 0, 1), (8448, 4096, 1024, 0, 0, 0), (8448, 1024, 4096, 0, 0, 0), (8448, 3104, 1024, 0, 0, 1),
             (8448, 1024, 1024, 0, 0, 0), (33792, 1024, 1024, 512, 1, 0), (33792, 1024, 1024, 1024, 0, 0), (33792, 512, 1024, 512, 1, 0),
             (8448, 4096, 512, 0, 0, 1), (8448, 512, 2048, 0, 0, 1), (8448, 512, 1024, 0, 0, 0), (8448, 1024, 512, 0, 0, 0)]
tn_shapes = [('ff1', 8448, 8192, 1024), ('ff2', 8448, 1024, 4096), ('qkv', 8448, 3104, 1024), ('attn out', 8448, 1024, 1024),
             ('skip / cross a<-a', 33792, 1024, 1024), ('cross a<-t', 33792, 1024, 512), ('cross t<-a', 33792, 512, 1024), ('cross t<-t', 33792, 512, 512),
             ('text ff1', 8448, 4096, 512), ('text ff2', 8448, 512, 2048), ('text qkv', 8448, 3104, 512), ('text out', 8448, 512, 1024)]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
res = dict(nt=[], tn=[])
for (M, N, K1, K2, rs, bs) in nt_shapes:
    torch.manual_seed(M + N + K1)
    a = torch.randn(M, K1, device=dev).to(bf16); a2 = torch.randn(M, K2, device=dev).to(bf16) if K2 else None
    b = (torch.randn(N, K1 + K2, device=dev) * 0.05).to(bf16)
    bias = torch.randn(N, device=dev) if bs else None
    resid = torch.randn(M, N, device=dev).to(bf16) if rs else None
    fl = 2.0 * M * N * (K1 + K2)
    row = dict(M=M, N=N, K1=K1, K2=K2, resid=rs, bias=bs)
    outs = {}
    for tag, f in (('staged', 0), ('direct', 64)):
        ops.gemm_flags = f
        outs[tag] = ops.gemm_nt(a, b, a2=a2, bias=bias, resid=resid).clone()
    row['same_bits'] = bool(torch.equal(outs['staged'], outs['direct']))
    out = torch.empty(M, N, device=dev, dtype=bf16)
    t = {'staged': [], 'direct': []}
    for rnd in range(3):
        for tag, f in (('staged', 0), ('direct', 64)):
            ops.gemm_flags = f
            t[tag].append(timeit(lambda: ops.gemm_nt(a, b, a2=a2, bias=bias, resid=resid, out=out)))
    for tag in t:
        ms = sorted(t[tag])[1]
        row[tag] = dict(us=round(ms * 1e3, 1), tf=round(fl / ms / 1e9, 1))
    ops.gemm_flags = 0
    res['nt'].append(row)
    print(row, flush=True)
for (tag, M, N, K) in tn_shapes:
    a = torch.randn(M, N, device=dev).to(bf16); b = torch.randn(M, K, device=dev).to(bf16)
    out = torch.zeros(N, K, device=dev)
    ms = sorted(timeit(lambda: ops.gemm_tn(a, b, out)) for _ in range(3))[1]
    ref = a[:2048].float().T @ b[:2048].float()
    out.zero_(); ops.gemm_tn(a[:2048], b[:2048], out)
    row = dict(tag=tag, M=M, N=N, K=K, us=round(ms * 1e3, 1), tf=round(2.0 * M * N * K / ms / 1e9, 1),
               rel_err_2048_rows=float((out - ref).norm() / ref.norm()))
    res['tn'].append(row)
    print(row, flush=True)
Path(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(res, open(ROOT / 'gpurun_out' / 'gemm_epilogue_ab.json', 'w'), indent=1)
