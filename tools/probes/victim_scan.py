"""which kernels change their results when the weight-gradient GEMM (glds + ds_read_b64_tr_b16) runs next to them on another stream?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16, f32 = torch.bfloat16, torch.float32
dev = 'cuda'
torch.manual_seed(0)
B, N, H, D = 4, 256, 8, 512
M = B * N
X = torch.randn(M, 4, D, device=dev).to(bf16)
params = [torch.ones(4, device=dev), torch.randn(4, 5, device=dev), torch.randn(D, 5, device=dev) * 0.03,
          torch.tensor(0.01, device=dev), torch.randn(D, device=dev) * 0.03, torch.tensor(0.01, device=dev), torch.zeros(D, device=dev)]
M1, b1, c1 = ops.hc_fwd(X, params)
y1 = torch.randn(M, D, device=dev).to(bf16)
M2, b2, c2 = ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1)
G = torch.randn(M, 4, D, device=dev).to(bf16); db = torch.randn(M, D, device=dev).to(bf16); y2 = torch.randn(M, D, device=dev).to(bf16)
xx = torch.randn(M, D, device=dev).to(bf16); gam = torch.randn(1, D, device=dev)
qkvg = torch.randn(M, 3 * D + H, device=dev).to(bf16)
cosb, sinb = ops.rotary_table(N, dev)
kmask = torch.ones(B, N, dtype=torch.uint8, device=dev)
st = ops.qkv_post_fwd(qkvg, B, H, N, cosb, sinb, None)
Og = ops.attn_fwd(st, kmask, 0.1, 1, 3).clone()
dOg = torch.randn(M, D, device=dev).to(bf16)
wnt = torch.randn(1024, D, device=dev).to(bf16)
cw = torch.randn(D, 31, device=dev); cb = torch.randn(D, device=dev)
Hh = torch.randn(M, 2 * 1024, device=dev).to(bf16)
a1 = torch.randn(1024, 1552, device=dev).to(bf16); b1 = torch.randn(1024, 512, device=dev).to(bf16)
out = torch.zeros(1552, 512, device=dev)
side = torch.cuda.Stream()
hgrads = [torch.zeros_like(p) for p in params]
def v_hc_bwd():
    grads = hgrads
    return ops.hc_bwd(G, xin=M1, yprev=y1, coef_prev=c1, dbin=db, ycur=y2, coef=c2, params=params, grads=grads)
def v_hc_bwd_depth():
    return (ops.hc_bwd(G, yprev=y2, coef_prev=c2)[1],)
def v_hc_fwd():
    return ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1)
def v_rms_fwd():
    return ops.rmsnorm_fwd(xx, gam, 1., M)
def v_rms_bwd():
    y, rn = ops.rmsnorm_fwd(xx, gam, 1., M)
    dg = torch.zeros(1, D, device=dev)
    return (ops.rmsnorm_bwd(db, xx, rn, gam, 1., M, dg),)
def v_attn_fwd():
    return (ops.attn_fwd(st, kmask, 0.1, 1, 3),)
def v_attn_bwd():
    return ops.attn_bwd(st, dOg, kmask, 0.1, 1, 3)
def v_nt():
    return (ops.gemm_nt(xx, wnt),)
def v_conv():
    pre, y = ops.dwconv_fwd(xx.view(B, N, D), None, cw, cb)
    return (pre, y)
def v_geglu():
    return (ops.geglu_fwd(Hh, 0.1, 1, 3) if hasattr(ops, 'geglu_fwd') else xx,)
victims = dict(hc_bwd=v_hc_bwd, hc_bwd_depth_only=v_hc_bwd_depth, hc_fwd=v_hc_fwd, rmsnorm_fwd=v_rms_fwd, rmsnorm_bwd=v_rms_bwd, attn_fwd=v_attn_fwd,
               attn_bwd=v_attn_bwd, gemm_nt=v_nt, dwconv_fwd=v_conv)
def run(fn, co):
    torch.cuda.synchronize()
    if co:
        with torch.cuda.stream(side):
            for _ in range(60):
                ops.gemm_tn(a1, b1, out)
    r = fn()
    torch.cuda.synchronize()
    return [t.clone() for t in r if torch.is_tensor(t)]
for name, fn in victims.items():
    ref = run(fn, False)
    res = {}
    for co in (False, True):
        bad = 0
        for _ in range(60):
            got = run(fn, co)
            bad += any(not torch.equal(a, b) for a, b in zip(got, ref))
        res[co] = bad
    print(f'{name:20s} alone: {res[False]:3d}/60 differ   next to the TN GEMM: {res[True]:3d}/60 differ', flush=True)
