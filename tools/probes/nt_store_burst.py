"""Where does the per-tile overhead of the 256 x 256 NT kernel go?  Shapes with many rounds of tiles (so that one-off costs
vanish), one-tile-per-workgroup kernel vs the persistent kernel, and two probes of the persistent kernel: K loops without
the C-tile stores (E2K_GEMM_PROBE_NO_STORE, wrong results) and workgroups started a quarter tile apart
(E2K_GEMM_PROBE_STAGGER).  us per round of 256 tiles = time / (tiles / 256).  -> gpurun_out/nt_store_burst.json"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
shapes = [(32768, 8192, 1024), (32768, 8192, 512), (16384, 8192, 4096), (33792, 1024, 1536)]
VARIANTS = (('t256', 128), ('persist', 128 | 64), ('persist_no_store', 128 | 64 | 2), ('persist_stagger', 128 | 64 | 512))
def timeit(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
rows = []
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev).to(bf16); b = (torch.randn(N, K, device=dev) * 0.05).to(bf16)
    out = torch.empty(M, N, device=dev, dtype=bf16)
    tiles = (M // 256) * (N // 256)
    row = dict(M=M, N=N, K=K, tiles=tiles, rounds=tiles / 256, k_tiles=K // 64)
    t = {tag: [] for tag, _ in VARIANTS}
    for rnd in range(3):
        for tag, f in VARIANTS:
            ops.gemm_flags = f
            t[tag].append(timeit(lambda: ops.gemm_nt(a, b, out=out)))
    for tag, _ in VARIANTS:
        ms = sorted(t[tag])[1]
        row[tag] = dict(us=round(ms * 1e3, 1), tf=round(2.0 * M * N * K / ms / 1e9, 1), us_per_round=round(ms * 1e3 / (tiles / 256), 2))
    ops.gemm_flags = 0
    rows.append(row)
    print(row, flush=True)
Path(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(rows, open(ROOT / 'gpurun_out' / 'nt_store_burst.json', 'w'), indent=1)
