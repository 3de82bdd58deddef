"""How does the 256x256 8-phase NT kernel do where the guide quotes its template (4096^3 / 8192^3, uniform random operands)?
Separates kernel quality from the shape effects of the model's GEMMs (K = 1024-2048, 33 x n tile grids)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
rows = []
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 8192, 1024), (8192, 8192, 2048), (8448, 8192, 1024), (16384, 4096, 1024), (8192, 4096, 4096), (33792, 1024, 1536), (8448, 4096, 1024), (8448, 3104, 1024)]:
    a = (torch.rand(M, K, device='cuda') * 2 - 1).to(bf16); b = (torch.rand(N, K, device='cuda') * 2 - 1).to(bf16)
    out = torch.empty(M, N, device='cuda', dtype=bf16)
    r = dict(M=M, N=N, K=K)
    for name, fl in (('t256', 128), ('t128', 256)):
        ops.gemm_flags = fl
        ms = timeit(lambda: ops.gemm_nt(a, b, out=out))
        r[name] = round(2.0 * M * N * K / ms / 1e9, 1)
    ms = timeit(lambda: torch.matmul(a, b.t(), out=out))
    r['hipblaslt'] = round(2.0 * M * N * K / ms / 1e9, 1)
    rows.append(r); print(r, flush=True)
ops.gemm_flags = 0
json.dump(rows, open(ROOT / 'gpurun_out' / 'r02_gemm_square.json', 'w'), indent=1)
