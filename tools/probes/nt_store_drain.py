"""is the NT epilogue's store drain a per-CU cost or the chip's HBM write rate?  One round of T = 32 .. 256 tiles of 256 x 256 (one
workgroup per CU on T CUs), bf16 output (128 KB per tile) against fp32 output (256 KB per tile) of the same product: the difference is
128 KB x T more stores and nothing else.  Per-CU bound: the difference does not depend on T.  Chip-wide (HBM write rate): it grows
linearly with T.    python tools/probes/nt_store_drain.py [out.json]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
dev = 'cuda'


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


ops.gemm_flags = 128                                   # the 256 x 256 8-phase kernel for every shape
rows = []
for K in (512, 1024, 2048):
    for T, (M, N) in ((32, (2048, 1024)), (64, (2048, 2048)), (128, (4096, 2048)), (192, (4096, 3072)), (256, (4096, 4096))):
        a = torch.randn(M, K, device=dev).to(bf16)
        b = torch.randn(N, K, device=dev).to(bf16)
        ob, of = torch.empty(M, N, device=dev, dtype=bf16), torch.empty(M, N, device=dev, dtype=f32)
        tb = min(timeit(lambda: ops.gemm_nt(a, b, out=ob)) for _ in range(3))
        tf = min(timeit(lambda: ops.gemm_nt(a, b, out=of)) for _ in range(3))
        extra = T * 128 * 1024
        rows.append(dict(K=K, tiles=T, bf16_us=round(tb, 2), fp32_us=round(tf, 2), extra_us=round(tf - tb, 2), extra_bytes=extra,
                         extra_GBps=round(extra / (tf - tb) / 1e3, 0) if tf > tb else None, tflops_bf16=round(2. * M * N * K / tb / 1e6, 0)))
        print(rows[-1], flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], 'w'), indent=1)
