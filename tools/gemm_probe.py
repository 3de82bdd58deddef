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
elif which == 'hc':
    D = 1024
    X = torch.randn(M, 4, D, device=dev).to(bf16)
    params = [torch.ones(4, device=dev), torch.randn(4, 5, device=dev), torch.randn(D, 5, device=dev) * 0.03,
              torch.tensor(0.01, device=dev), torch.randn(D, device=dev) * 0.03, torch.tensor(0.01, device=dev),
              torch.zeros(D, device=dev)]
    grads = [torch.zeros_like(p) for p in params]
    M1, b1, c1 = ops.hc_fwd(X, params)
    y1 = torch.randn(M, D, device=dev).to(bf16)
    M2, b2, c2 = ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1)
    G = torch.randn(M, 4, D, device=dev).to(bf16); db = torch.randn(M, D, device=dev).to(bf16); y2 = torch.randn(M, D, device=dev).to(bf16)
    for _ in range(4):
        ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1)
        ops.hc_bwd(G, xin=M1, yprev=y1, coef_prev=c1, dbin=db, ycur=y2, coef=c2, params=params, grads=grads)
torch.cuda.synchronize()
