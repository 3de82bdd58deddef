"""Token-split weight gradients: GEMM + tn_reduce_kernel (two launches) against the in-kernel finish
(e2k_gemm_tn_self_reduce_bf16: the last workgroup of a tile to arrive sums the partial tiles) on the cfg3 shapes --
bit comparison, 50 repeats for reproducibility (arrival order must not matter), back-to-back timing with HIP events.
-> gpurun_out/tn_self_reduce.json        (UNMEASURED so far: written when no GPU time was left in round 2)"""
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


def timeit(fn, iters=20):
    for _ in range(3):
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
for (M, N, K, tag) in shapes:
    a = torch.randn(M, N, device=dev).to(bf16)
    b = torch.randn(M, K, device=dev).to(bf16)
    row = dict(tag=tag, M=M, N=N, K=K, splits=ops.lib().e2k_query_gemm_tn_splits_mode(M, N, K, 0, ops.tn_mode))
    fl = 2.0 * M * N * K
    res = {}
    for fused in (False, True):
        ops.tn_self_reduce = fused
        out = torch.zeros(N, K, device=dev)
        ops.gemm_tn(a, b, out)
        res[fused] = out.clone()
        if fused:
            bad = 0
            for _ in range(50):
                out.zero_()
                ops.gemm_tn(a, b, out)
                bad += int(not torch.equal(out, res[True]))
            row['irreproducible_of_50'] = bad
        ms = timeit(lambda: ops.gemm_tn(a, b, out))
        row['self_reduce' if fused else 'two_launches'] = dict(us=round(ms * 1e3, 1), tf=round(fl / ms / 1e9, 1))
    ops.tn_self_reduce = False
    row['mismatching_elements'] = int((res[True] != res[False]).sum())
    rows.append(row)
    print(row, flush=True)
Path(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(rows, open(ROOT / 'gpurun_out' / 'tn_self_reduce.json', 'w'), indent=1)
