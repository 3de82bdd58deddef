"""FeedForward GEMM1 + GEGLU: one launch with the GEGLU as the GEMM's epilogue (e2k_gemm_nt_geglu_bf16) against the two
launches (e2k_gemm_nt_bf16 + e2k_geglu_fwd), on the cfg3 audio / text shapes, training form (H stored) and inference
form (H not stored), with and without dropout.  Back-to-back launches timed with HIP events (the cadence of a plan
replay).  -> gpurun_out/geglu_fused.json        (UNMEASURED so far: written when no GPU time was left in round 2)"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
shapes = [(8448, 4096, 1024), (8448, 2048, 512), (33792, 4096, 1024)]          # (tokens, F, D): cfg3 audio, cfg3 text, 4 x the tokens


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


rows = []
for (M, F, K) in shapes:
    a = (torch.randn(M, K, device=dev) * 0.5).to(bf16)
    w1 = (torch.randn(2 * F, K, device=dev) * 0.05).to(bf16)
    b1 = torch.randn(2 * F, device=dev)
    fl = 2.0 * M * 2 * F * K
    row = dict(M=M, F=F, K=K)
    for p in (0.0, 0.1):
        H0 = ops.gemm_nt(a, w1, bias=b1)
        act0 = ops.geglu_fwd(H0, p, 11, 3)
        H1, act1 = ops.gemm_nt_geglu(a, w1, b1, p, 11, 3)
        row[f'mismatch_p{p}'] = dict(H=int((H0 != H1).sum()), act=int((act0 != act1).sum()))
        t_pair = timeit(lambda: ops.geglu_fwd(ops.gemm_nt(a, w1, bias=b1), p, 11, 3))
        t_gemm = timeit(lambda: ops.gemm_nt(a, w1, bias=b1))
        t_fused = timeit(lambda: ops.gemm_nt_geglu(a, w1, b1, p, 11, 3))
        t_inf = timeit(lambda: ops.gemm_nt_geglu(a, w1, b1, p, 11, 3, want_h=False))
        row[f'p{p}'] = dict(pair_us=round(t_pair * 1e3, 1), gemm_alone_us=round(t_gemm * 1e3, 1), fused_us=round(t_fused * 1e3, 1),
                            fused_no_H_us=round(t_inf * 1e3, 1), fused_tf=round(fl / t_fused / 1e9, 1), gemm_alone_tf=round(fl / t_gemm / 1e9, 1))
    rows.append(row)
    print(row, flush=True)
Path(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(rows, open(ROOT / 'gpurun_out' / 'geglu_fused.json', 'w'), indent=1)
