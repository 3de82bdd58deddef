"""Host logic check of the row / element kernels against plain torch (fp32 autograd)."""
import pytest
import torch
import torch.nn.functional as F

bf16 = torch.bfloat16


def rel(a, b):
    a, b = a.cpu(), b.cpu()
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-20)).item()


@pytest.mark.parametrize('D,nb', [(128, 1), (256, 3), (512, 1), (768, 2), (1024, 2), (1536, 1), (2048, 2)])
def test_rmsnorm(dev, D, nb):
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    rpb = 13
    M = rpb * nb - (2 if nb > 1 else 0)
    x = torch.randn(M, D).to(bf16)
    gamma = torch.randn(nb, D) * 0.3
    off = 1.0 if nb > 1 else 0.0
    dy = torch.randn(M, D).to(bf16)
    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    idx = torch.arange(M) // rpb
    yr = F.normalize(xr, dim=-1) * D ** 0.5 * (gr[idx] + off)
    yr.backward(dy.float())
    x, gamma, dy = x.to(dev), gamma.to(dev), dy.to(dev)
    y, rn = ops.rmsnorm_fwd(x, gamma, off, rpb)
    assert rel(y, yr) < 1e-2
    dg = torch.zeros(nb, D, device=dev)
    dx = ops.rmsnorm_bwd(dy, x, rn, gamma, off, rpb, dg)
    assert rel(dx, xr.grad) < 1e-2
    assert rel(dg, gr.grad) < 1e-2


@pytest.mark.parametrize('D', [256, 512, 1024])
def test_gate_bwd(dev, D):
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    rpb, nb = 11, 3
    M = rpb * nb
    ao = torch.randn(M, D)
    g = torch.rand(nb, D) * 0.5 + 0.1
    idx = torch.arange(M) // rpb
    y = (ao * g[idx]).to(bf16)
    dy = torch.randn(M, D).to(bf16)
    gsum = torch.zeros(nb, D, device=dev)
    dao = ops.gate_bwd(dy.to(dev), y.to(dev), g.to(dev), gsum, rpb)
    assert rel(dao, dy.float() * g[idx]) < 1e-2
    ref = torch.zeros(nb, D).index_add_(0, idx, dy.float() * y.float())
    assert rel(gsum, ref) < 1e-4


def test_geglu(dev):
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    M, Fd = 19, 64
    H = torch.randn(M, 2 * Fd).to(bf16)
    da = torch.randn(M, Fd).to(bf16)
    Hr = H.float().requires_grad_(True)
    u, gt = Hr.chunk(2, dim=-1)
    a = u * F.gelu(gt)
    a.backward(da.float())
    out = ops.geglu_fwd(H.to(dev))
    assert rel(out, a) < 1e-2
    dH = ops.geglu_bwd(da.to(dev), H.to(dev))
    assert rel(dH, Hr.grad) < 1e-2


def test_colsum_cast(dev):
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    x = torch.randn(77, 130).to(bf16)
    out = torch.ones(130, device=dev)
    ops.colsum(x.to(dev), out)
    assert rel(out, 1 + x.float().sum(0)) < 1e-5
    src = torch.randn(1003)
    dst = torch.empty(1003, dtype=bf16, device=dev)
    # 16-byte alignment of the test buffers is what torch gives us
    ops.cast_bf16(src.to(dev), dst)
    assert torch.equal(dst.cpu(), src.to(bf16))
    w = torch.randn(70, 45)
    wt = torch.empty(45, 70, dtype=bf16, device=dev)
    ops.cast_transpose_bf16(w.to(dev), wt)
    assert torch.equal(wt.cpu(), w.t().to(bf16))


def test_cast_transpose_batch(dev):
    """every transposed bf16 shadow of a module in one launch: ragged shapes, padded destination rows, matrices at
    8-element-aligned offsets of one flat buffer (the backbone's layout), untouched gaps"""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    shapes = [(70, 45, 72), (128, 64, 128), (8, 200, 8), (3104, 256, 3136), (64, 1, 64), (130, 136, 136)]     # (R, C, ldd)
    flat, rows, soff, doff, blk = [], [], 0, 0, 0
    for R, C, ldd in shapes:
        flat.append(torch.randn(R * C))
        flat.append(torch.zeros((-R * C) % 8))
        rows.append([soff, doff, R, C, ldd, blk])
        soff += R * C + (-R * C) % 8
        doff += (C * ldd + 7) // 8 * 8
        blk += ((R + 63) // 64) * ((C + 63) // 64)
    flat = torch.cat(flat)
    flatT = torch.full((doff,), -7.0, dtype=bf16, device=dev)
    ops.cast_transpose_batch(flat.to(dev), flatT, torch.tensor(rows, dtype=torch.int64, device=dev), blk)
    got = flatT.cpu()
    for (R, C, ldd), (so, do, *_rest) in zip(shapes, rows):
        w = flat[so:so + R * C].view(R, C)
        t = got[do:do + C * ldd].view(C, ldd)
        assert torch.equal(t[:, :R], w.t().to(bf16)), (R, C)
        assert bool((t[:, R:] == -7.0).all())                 # pad columns of the destination rows are left alone


# (frames, kernel, mask): 75 frames = two tiles with a ragged tail; 200 frames = several tiles per workgroup in the backward
@pytest.mark.parametrize('N,ks,use_mask', [(75, 31, True), (75, 7, False), (200, 31, True), (64, 15, False), (5, 3, False)])
@pytest.mark.parametrize('split', [False, True])
def test_dwconv(dev, N, ks, use_mask, split):
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    B, C = 2, 128
    x = torch.randn(B, N, C).to(bf16)
    w = torch.randn(C, 1, ks) * 0.3
    bias = torch.randn(C) * 0.1
    mask = None
    if use_mask:
        lens = torch.tensor([N, N - 25])
        mask = torch.arange(N)[None] < lens[:, None]
    dy = torch.randn(B, N, C).to(bf16)
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    xm = xr if mask is None else torch.where(mask[..., None], xr, torch.zeros_like(xr))
    pre_r = F.conv1d(xm.transpose(1, 2), wr, br, padding=ks // 2, groups=C).transpose(1, 2)
    yr = F.silu(pre_r)
    if mask is not None:
        yr = torch.where(mask[..., None], yr, torch.zeros_like(yr))
    yr.backward(dy.float())
    xd, wd, bd, md = x.to(dev), w.to(dev), bias.to(dev), (None if mask is None else mask.to(dev))
    pre, y = ops.dwconv_fwd(xd, md, wd, bd)
    assert rel(y, yr) < 1e-2 and rel(pre, pre_r.detach() if mask is None else torch.where(mask[..., None], pre_r.detach(), pre.cpu().float())) < 1e-2
    none, y2 = ops.dwconv_fwd(xd, md, wd, bd, need_pre=False)          # no-grad form: the pre-activation is not written
    assert none is None and torch.equal(y2.cpu(), y.cpu())
    dw, db = torch.zeros_like(wd), torch.zeros_like(bd)
    ops.dwconv_bwd_workspace = not split          # (True: workspace + reduce pass, the default; False: global atomics)
    try:
        dx = ops.dwconv_bwd(dy.to(dev), pre, xd, md, wd, dw, db)
    finally:
        ops.dwconv_bwd_workspace = True
    assert rel(dx, xr.grad) < 2e-2, rel(dx, xr.grad)
    assert rel(dw, wr.grad) < 2e-2, rel(dw, wr.grad)
    assert rel(db, br.grad) < 2e-2


def test_geglu_dropout(dev):
    from e2_tts_pytorch_amd import ops
    from oracle.dropout_hash import geglu_dropout_mask
    torch.manual_seed(0)
    M, Fd, p, seed, sid = 33, 64, 0.1, 777, 9
    H = torch.randn(M, 2 * Fd).to(bf16)
    da = torch.randn(M, Fd).to(bf16)
    mask = geglu_dropout_mask(seed, sid, M, Fd, p)
    assert 0.03 < (mask == 0).float().mean().item() < 0.2
    Hr = H.float().requires_grad_(True)
    u, gt = Hr.chunk(2, dim=-1)
    a = u * F.gelu(gt) * mask
    a.backward(da.float())
    out = ops.geglu_fwd(H.to(dev), p, seed, sid)
    assert torch.equal(out.cpu() == 0, (a == 0)) or rel(out, a) < 1e-2
    assert rel(out, a) < 1e-2
    dH = ops.geglu_bwd(da.to(dev), H.to(dev), p, seed, sid)
    assert rel(dH, Hr.grad) < 1e-2
