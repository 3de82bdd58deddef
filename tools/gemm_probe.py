"""run one kernel family a few times (for rocprofv3 --pmc passes)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else 'nt'
bf16 = torch.bfloat16
dev = 'cuda'
M = 8448
if which == 'nt':
    a = torch.randn(M, 1024, device=dev).to(bf16); b = torch.randn(8192, 1024, device=dev).to(bf16)
    for _ in range(5):
        ops.gemm_nt(a, b)
elif which == 'tn':
    a = torch.randn(M, 8192, device=dev).to(bf16); b = torch.randn(M, 1024, device=dev).to(bf16)
    out = torch.zeros(8192, 1024, device=dev)
    for _ in range(5):
        ops.gemm_tn(a, b, out)
elif which == 'attn':
    B, H, N = 8, 16, 1056
    qkvg = torch.randn(B * N, 3 * 1024 + 32, device=dev).to(bf16)
    cosb, sinb = ops.rotary_table(N, dev)
    st = ops.qkv_post_fwd(qkvg, B, H, N, cosb, sinb, None)
    kmask = torch.zeros(B, st.Npad, dtype=torch.uint8, device=dev); kmask[:, :N] = 1
    dOg = torch.randn(B * N, 1024, device=dev).to(bf16)
    for _ in range(3):
        ops.attn_fwd(st, kmask, 0.1, 1, 3)
        ops.attn_bwd(st, dOg, kmask, 0.1, 1, 3)
torch.cuda.synchronize()
