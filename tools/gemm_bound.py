"""NT GEMM bottleneck probes: full kernel vs K loop without loads (LDS+MFMA side) vs without math (load side)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'

def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

for (M, N, K) in [(8448, 8192, 1024), (8448, 1024, 4096), (8448, 1024, 1024), (33792, 1024, 2048), (33792, 512, 1536), (8448, 3104, 1024), (8448, 3104, 512), (8448, 1024, 8192), (8192, 1024, 4096), (8192, 8192, 1024), (8448, 4096, 512), (8448, 512, 2048)]:
    a = torch.randn(M, K, device=dev).to(bf16); b = torch.randn(N, K, device=dev).to(bf16)
    row = {}
    for name, fl in (('big', 64), ('big_nosplit', 64 | 16), ('t128', 0), ('t128_nosplit', 16), ('no_loads', 4 | 16), ('no_math', 8 | 16), ('neither', 12 | 16)):
        ops.gemm_flags = fl
        ms = timeit(lambda: ops.gemm_nt(a, b))
        row[name] = round(ms * 1e3, 1)
    ops.gemm_flags = 0
    print(f'{M}x{N}x{K}', row, 'TF big', round(2 * M * N * K / row['big'] / 1e6, 1), 'TF t128', round(2 * M * N * K / row['t128'] / 1e6, 1), flush=True)
