"""one gemm_nt launch per distinct shape of tools/nt_shapes_cfg3.json (written by tools/nt_shapes.py), in file order (run under rocprofv3 --pmc)"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
shapes = json.load(open(ROOT / 'tools' / 'nt_shapes_cfg3.json'))
for s in shapes:
    M, N, K1, K2 = s['M'], s['N'], s['K1'], s['K2']
    a = torch.randn(M, K1, device='cuda').to(bf16)
    a2 = torch.randn(M, K2, device='cuda').to(bf16) if K2 else None
    b = torch.randn(N, K1 + K2, device='cuda').to(bf16)
    r = torch.randn(M, N, device='cuda').to(bf16) if s['resid'] else None
    torch.cuda.synchronize()
    ops.gemm_nt(a, b, a2=a2, resid=r, out_dtype=torch.float32 if s['out_f32'] else bf16)
    torch.cuda.synchronize()
