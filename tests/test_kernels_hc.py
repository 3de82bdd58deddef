"""Host logic check of the fused hyper-connection kernels against the oracle's HyperConnections (+autograd)."""
import pytest
import torch

from oracle.e2tts_oracle import HyperConnections

bf16 = torch.bfloat16


def _mk_hc(D, seed):
    import random
    random.seed(seed)
    torch.manual_seed(seed)
    hc = HyperConnections(4, dim=D)
    with torch.no_grad():      # make every term matter
        hc.dynamic_alpha_fn.normal_(0, D ** -0.5)
        hc.dynamic_beta_fn.normal_(0, D ** -0.5)
        hc.dynamic_alpha_scale.fill_(0.7)
        hc.dynamic_beta_scale.fill_(0.5)
        hc.norm.gamma.normal_(0, 0.3)
        hc.static_alpha.add_(torch.randn_like(hc.static_alpha) * 0.2)
        hc.static_beta.add_(torch.randn_like(hc.static_beta) * 0.2)
    return hc


def _params(hc):
    return [hc.static_beta, hc.static_alpha, hc.dynamic_alpha_fn, hc.dynamic_alpha_scale, hc.dynamic_beta_fn,
            hc.dynamic_beta_scale, hc.norm.gamma]


def _to_ref_layout(x):        # (Mtok,4,D) token-major -> '(b s) n d' with b = 1
    return x.permute(1, 0, 2).contiguous()           # (4, Mtok, D)  == (b s) n d with b=1


def _from_ref_layout(x):
    return x.permute(1, 0, 2).contiguous()


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-20)).item()


def _ok(got, want, D=1024):
    if got.numel() == 1:        # heavily cancelling sums of bf16-rounded terms: absolute slack
        return abs(got.item() - want.item()) <= 0.15 * abs(want.item()) + 1.5
    if got.numel() <= 32 and D > 1024:      # the 4 / 20 static gradients are sums over D * tokens bf16 products: noise ~ sqrt(D)
        return rel(got, want) < 5e-2
    return rel(got, want) < 3e-2


@pytest.mark.parametrize('D', [128, 256, 512, 768, 1024, 1536, 2048])
def test_hc_chain(dev, D):
    from e2_tts_pytorch_amd import ops
    Mtok = 37
    hc1, hc2 = _mk_hc(D, 1), _mk_hc(D, 2)
    torch.manual_seed(3)
    X = torch.randn(Mtok, 4, D).to(bf16)
    to = lambda t: t.to(dev)
    Wl = torch.randn(Mtok, 4, D)
    f1 = lambda t: torch.tanh(t) * 1.5
    f2 = lambda t: torch.sin(t) + 0.1 * t

    # ---- oracle chain (fp32, autograd)
    Xr = X.float().clone().requires_grad_(True)
    xr = _to_ref_layout(Xr)
    b1, add1 = hc1(xr)
    y1r = f1(b1)
    x2 = add1(y1r)
    b2, add2 = hc2(x2)
    y2r = f2(b2)
    x3 = add2(y2r)
    loss = (_from_ref_layout(x3) * Wl).sum()
    loss.backward()

    # ---- kernels: width1 ; depth1+width2 ; depth2 (materialise)
    p1 = [to(t.detach()) for t in _params(hc1)]
    p2 = [to(t.detach()) for t in _params(hc2)]
    Xd = to(X)
    M1, bin1, c1 = ops.hc_fwd(Xd, p1)
    y1 = f1(bin1.float()).to(bf16)
    M2, bin2, c2 = ops.hc_fwd(M1, p2, yprev=y1, coef_prev=c1)
    y2 = f2(bin2.float()).to(bf16)
    X3, _, _ = ops.hc_fwd(M2, None, yprev=y2, coef_prev=c2, width=False)
    assert rel(bin1.cpu(), b1[0]) < 2e-2 and rel(bin2.cpu(), b2[0]) < 2e-2
    assert rel(X3.cpu(), _from_ref_layout(x3)) < 2e-2

    # ---- backward
    g1 = [torch.zeros_like(t) for t in p1]
    g2 = [torch.zeros_like(t) for t in p2]
    dX3 = to(Wl.to(bf16))
    _, dy2 = ops.hc_bwd(dX3, yprev=y2, coef_prev=c2)                       # depth2 only
    b2k = bin2.float().requires_grad_(True)
    f2(b2k).backward(dy2.float())
    dbin2 = b2k.grad.to(bf16)
    dM1, dy1 = ops.hc_bwd(dX3, xin=M1, yprev=y1, coef_prev=c1, dbin=dbin2, ycur=y2, coef=c2, params=p2, grads=g2)
    b1k = bin1.float().requires_grad_(True)
    f1(b1k).backward(dy1.float())
    dbin1 = b1k.grad.to(bf16)
    dX, _ = ops.hc_bwd(dM1, xin=Xd, dbin=dbin1, ycur=y1, coef=c1, params=p1, grads=g1)
    assert rel(dX.cpu(), Xr.grad) < 3e-2, rel(dX.cpu(), Xr.grad)
    for name, gk, pr in zip(ops.HC_PARAM_NAMES, g2, _params(hc2)):
        assert _ok(gk.cpu(), pr.grad, D), (name, 2, gk, pr.grad)
    for name, gk, pr in zip(ops.HC_PARAM_NAMES, g1, _params(hc1)):
        assert _ok(gk.cpu(), pr.grad, D), (name, 1, gk, pr.grad)
