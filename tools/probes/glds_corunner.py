"""is the LDS-DMA GEMM itself still right when another kernel (hc_bwd, 25 KB LDS per workgroup) shares its CUs?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
torch.manual_seed(0)
M, D = 4096, 512
X = torch.randn(M, 4, D, device=dev).to(bf16)
params = [torch.ones(4, device=dev), torch.randn(4, 5, device=dev), torch.randn(D, 5, device=dev) * 0.03,
          torch.tensor(0.01, device=dev), torch.randn(D, device=dev) * 0.03, torch.tensor(0.01, device=dev), torch.zeros(D, device=dev)]
M1, b1, c1 = ops.hc_fwd(X, params)
y1 = torch.randn(M, D, device=dev).to(bf16)
M2, b2, c2 = ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1)
G = torch.randn(M, 4, D, device=dev).to(bf16); db = torch.randn(M, D, device=dev).to(bf16); y2 = torch.randn(M, D, device=dev).to(bf16)
grads = [torch.zeros_like(p) for p in params]
an = torch.randn(2048, 1024, device=dev).to(bf16); wn = torch.randn(2048, 1024, device=dev).to(bf16)
side = torch.cuda.Stream()
def gemm():
    return ops.gemm_nt(an, wn)
ref = gemm(); torch.cuda.synchronize()
for co in (False, True):
    bad = 0
    for t in range(100):
        torch.cuda.synchronize()
        if co:
            with torch.cuda.stream(side):
                for _ in range(8):
                    ops.hc_bwd(G, xin=M1, yprev=y1, coef_prev=c1, dbin=db, ycur=y2, coef=c2, params=params, grads=grads)
        got = gemm()
        torch.cuda.synchronize()
        if not torch.equal(got, ref):
            bad += 1
            if bad <= 2:
                d = (got != ref)
                print('   rows', d.any(1).sum().item(), 'cols', d.any(0).sum().item(), 'max abs', float((got.float() - ref.float()).abs().max()))
    print('GEMM (LDS-DMA) next to hc_bwd' if co else 'GEMM alone', bad, 'of 100 differ', flush=True)
