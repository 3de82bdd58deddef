"""Per-kernel timings at the cfg3 shapes (dim 1024, heads 16, B 8, N 1056) on one MI355X -> gpurun_out/microbench.json.
HIP events on the current stream; achieved TFLOP/s (MFMA kernels) or GB/s of algorithmic bytes (HBM-bound kernels)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch  # noqa: E402

from e2_tts_pytorch_amd import ops  # noqa: E402

bf16, f32 = torch.bfloat16, torch.float32
dev = 'cuda'
res = []


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rec(name, ms, flops=None, bytes_=None, **kw):
    r = dict(name=name, ms=round(ms, 4), **kw)
    if flops:
        r['tflops'] = round(flops / ms / 1e9, 1)
    if bytes_:
        r['gbps'] = round(bytes_ / ms / 1e6, 1)
    res.append(r)
    print(r, flush=True)


def rnd(*shape, dtype=bf16):
    return torch.randn(*shape, device=dev, dtype=torch.float32).to(dtype)


B, H, T, R = 8, 16, 1024, 32
N = T + R
M = B * N
which = sys.argv[1:] or ['gemm', 'tn', 'attn', 'hc', 'ew']

if 'gemm' in which:
    for (m, n, k1, k2) in [(M, 3104, 1024, 0), (M, 8192, 1024, 0), (M, 1024, 4096, 0), (M, 1024, 1024, 0),
                           (4 * M, 1024, 1024, 1024), (4 * M, 1024, 1024, 512), (4 * M, 512, 1024, 512),
                           (M, 3104, 512, 0), (M, 4096, 512, 0), (M, 512, 2048, 0), (8, 98304, 1024, 0)]:
        a, b = rnd(m, k1), rnd(n, k1 + k2)
        a2 = rnd(m, k2) if k2 else None
        ms = timeit(lambda: ops.gemm_nt(a, b, a2=a2))
        rec(f'gemm_nt {m}x{n}x{k1}+{k2}', ms, flops=2.0 * m * n * (k1 + k2))
        if k2 == 0:
            ms = timeit(lambda: torch.matmul(a, b.t()))
            rec(f'  torch.matmul (hipBLASLt) {m}x{n}x{k1}', ms, flops=2.0 * m * n * k1)
if 'tn' in which:
    for (m, n, k) in [(M, 8192, 1024), (M, 1024, 4096), (M, 3104, 1024), (M, 1024, 1024), (4 * M, 1024, 1536), (4 * M, 512, 1536), (4 * M, 1024, 2048), (M, 3104, 512), (M, 4096, 512), (M, 512, 2048)]:
        a, b = rnd(m, n), rnd(m, k)
        out = torch.zeros(n, k, device=dev)
        for tr in (1,):
            ms = timeit(lambda: ops.gemm_tn(a, b, out, use_tr=tr))
            rec(f'gemm_tn use_tr={tr} M{m} N{n} K{k}', ms, flops=2.0 * m * n * k)
        ms = timeit(lambda: torch.matmul(a.t(), b))
        rec(f'  torch.matmul a.T@b M{m} N{n} K{k}', ms, flops=2.0 * m * n * k)
if 'attn' in which:
    I = H * 64
    cols = 3 * I + 2 * H
    qkvg = rnd(M, cols)
    cosb, sinb = ops.rotary_table(N, dev)
    vfirst = rnd(B, H, N, 64)
    st = None

    def post():
        global st
        st = ops.qkv_post_fwd(qkvg, B, H, N, cosb, sinb, vfirst)
    ms = timeit(post)
    rec('qkv_post_fwd', ms, bytes_=M * cols * 2 + 7 * M * I * 2)
    kmask = torch.zeros(B, st.Npad, dtype=torch.uint8, device=dev)
    kmask[:, :N] = 1
    af = 4.0 * B * H * N * N * 64
    ops.attn_share_dropmask = False
    for pd in (0.0, 0.1):
        ms = timeit(lambda: ops.attn_fwd(st, kmask, pd, 1, 3))
        rec(f'attn_fwd p_drop={pd}', ms, flops=af)
        dOg = rnd(M, I)
        ms = timeit(lambda: ops.attn_bwd(st, dOg, kmask, pd, 1, 3))
        rec(f'attn_bwd (prep+dq+dkv) p_drop={pd}', ms, flops=2.5 * af, note='tflops counts the algorithmic 5 matmuls; the kernels execute 7')
    ops.attn_share_dropmask = True                 # default: dropout keep masks handed from the forward to the backward
    ms = timeit(lambda: ops.attn_fwd(st, kmask, 0.1, 1, 3))
    rec('attn_fwd p_drop=0.1, writes shared dropout masks', ms, flops=af)
    ms = timeit(lambda: ops.attn_bwd(st, dOg, kmask, 0.1, 1, 3))
    rec('attn_bwd p_drop=0.1, reads shared dropout masks', ms, flops=2.5 * af)
    q, k, v = (rnd(B, H, N, 64) for _ in range(3))
    ms = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
    rec('  torch SDPA fwd (no softclamp/gate)', ms, flops=af)
if 'hc' in which:
    for D in (1024, 512):
        X = rnd(M, 4, D)
        params = [torch.ones(4, device=dev), torch.randn(4, 5, device=dev), torch.randn(D, 5, device=dev) * 0.03,
                  torch.tensor(0.01, device=dev), torch.randn(D, device=dev) * 0.03, torch.tensor(0.01, device=dev),
                  torch.zeros(D, device=dev)]
        grads = [torch.zeros_like(p) for p in params]
        M1, b1, c1 = ops.hc_fwd(X, params)
        y1 = rnd(M, D)
        ms = timeit(lambda: ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1))
        rec(f'hc_fwd depth+width D={D}', ms, bytes_=M * D * 2 * 10)
        M2, b2, c2 = ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1)
        G, db, y2 = rnd(M, 4, D), rnd(M, D), rnd(M, D)
        ms = timeit(lambda: ops.hc_bwd(G, xin=M1, yprev=y1, coef_prev=c1, dbin=db, ycur=y2, coef=c2, params=params, grads=grads))
        rec(f'hc_bwd depth+width D={D}', ms, bytes_=M * D * 2 * 16)
        ms = timeit(lambda: ops.hc_fwd(M2, None, yprev=y2, coef_prev=c2, width=False))
        rec(f'hc_fwd materialise D={D}', ms, bytes_=M * D * 2 * 9)
if 'ew' in which:
    D = 1024
    x = rnd(M, D)
    gam = torch.randn(B, D, device=dev)
    ms = timeit(lambda: ops.rmsnorm_fwd(x, gam, 1.0, N))
    rec('rmsnorm_fwd', ms, bytes_=M * D * 4)
    y, rn = ops.rmsnorm_fwd(x, gam, 1.0, N)
    dg = torch.zeros_like(gam)
    dy = rnd(M, D)
    ms = timeit(lambda: ops.rmsnorm_bwd(dy, x, rn, gam, 1.0, N, dg))
    rec('rmsnorm_bwd', ms, bytes_=M * D * 6)
    ms = timeit(lambda: ops.gate_bwd(dy, y, gam, dg, N))
    rec('gate_bwd', ms, bytes_=M * D * 6)
    Hh = rnd(M, 8192)
    for pd in (0.0, 0.1):
        ms = timeit(lambda: ops.geglu_fwd(Hh, pd, 1, 2))
        rec(f'geglu_fwd p_drop={pd}', ms, bytes_=M * 12288 * 2)
        da = rnd(M, 4096)
        ms = timeit(lambda: ops.geglu_bwd(da, Hh, pd, 1, 2))
        rec(f'geglu_bwd p_drop={pd}', ms, bytes_=M * (4096 + 8192 + 8192) * 2)
    w, bias = torch.randn(D, 31, device=dev), torch.randn(D, device=dev)
    xc = rnd(B, N, D)
    ms = timeit(lambda: ops.dwconv_fwd(xc, None, w, bias))
    rec('dwconv_fwd', ms, bytes_=M * D * 6)
    pre, yy = ops.dwconv_fwd(xc, None, w, bias)
    dw, dbi = torch.zeros_like(w), torch.zeros_like(bias)
    ms = timeit(lambda: ops.dwconv_bwd(xc, pre, xc, None, w, dw, dbi))
    rec('dwconv_bwd', ms, bytes_=M * D * 8)
    ops.dwconv_bwd_workspace = False
    ms = timeit(lambda: ops.dwconv_bwd(xc, pre, xc, None, w, dw, dbi))
    rec('dwconv_bwd with global atomics instead of the workspace pass', ms, bytes_=M * D * 8)
    ops.dwconv_bwd_workspace = True
    out = torch.zeros(8192, device=dev)
    ms = timeit(lambda: ops.colsum(Hh, out))
    rec('colsum M x 8192', ms, bytes_=M * 8192 * 2)
    src = torch.randn(100_000_000, device=dev)
    dst = torch.empty(100_000_000, device=dev, dtype=bf16)
    ms = timeit(lambda: ops.cast_bf16(src, dst))
    rec('cast_bf16 100M', ms, bytes_=100e6 * 6)
    wsrc = torch.randn(8192, 1024, device=dev)
    wdst = torch.empty(1024, 8192, device=dev, dtype=bf16)
    ms = timeit(lambda: ops.cast_transpose_bf16(wsrc, wdst))
    rec('cast_transpose 8192x1024', ms, bytes_=8192 * 1024 * 6)

out_dir = ROOT / 'gpurun_out'
out_dir.mkdir(exist_ok=True)
(out_dir / 'microbench.json').write_text(json.dumps(res, indent=1))
