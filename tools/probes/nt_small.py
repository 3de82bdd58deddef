"""NT GEMM on the narrow / short-K cfg3 shapes: default selection vs forced 256 x 256 vs no remainder split, back-to-back
launches timed with HIP events (the launch cadence of a plan replay).  -> gpurun_out/nt_small.json"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
shapes = [(8448, 1024, 1024), (8448, 512, 1024), (8448, 1024, 512), (8448, 2048, 512), (8448, 512, 2048), (8448, 512, 4096), (8448, 4096, 512),
          (8448, 1024, 4096), (8448, 4096, 1024), (8448, 3104, 1024), (8448, 3104, 512), (8448, 1024, 3136), (8448, 512, 3136),
          (8448, 8192, 1024), (8448, 1024, 8192), (33792, 1024, 1024), (33792, 512, 1536), (33792, 1024, 1536), (33792, 1024, 2048)]
def timeit(fn, iters=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
rows = []
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(bf16); b = torch.randn(N, K, device=dev).to(bf16)
    out = torch.empty(M, N, device=dev, dtype=bf16)
    fl = 2.0 * M * N * K
    row = dict(M=M, N=N, K=K)
    for tag, fl_ in (('default', 0), ('t256', 128), ('no_t256', 256), ('no_split', int(sys.argv[1]) if len(sys.argv) > 1 else 0)):
        if tag == 'no_split' and not fl_:
            continue
        ops.gemm_flags = fl_
        ms = timeit(lambda: ops.gemm_nt(a, b, out=out))
        row[tag] = dict(us=round(ms * 1e3, 1), tf=round(fl / ms / 1e9, 1))
    ops.gemm_flags = 0
    t0 = timeit(lambda: torch.matmul(a, b.T, out=out))
    row['hipblaslt'] = dict(us=round(t0 * 1e3, 1), tf=round(fl / t0 / 1e9, 1))
    rows.append(row)
    print(row, flush=True)
Path(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(rows, open(ROOT / 'gpurun_out' / 'nt_small.json', 'w'), indent=1)
