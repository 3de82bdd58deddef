"""Host logic check of the whole hand-scheduled backbone (forward + backward) against the oracle Transformer."""
import random

import pytest
import torch

from conftest import gpu_shapes

from oracle import e2tts_oracle as O

bf16 = torch.bfloat16


def rel(a, b):
    a, b = a.cpu(), b.cpu()
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-20)).item()


def rel2(a, b):
    """relative L2 error: bf16 rounding noise through the whole backbone sits at 1-3 %, a wrong term at >= 10 %"""
    a, b = a.cpu(), b.cpu()
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


def randomize(model, seed=0):
    """make every zero-initialised path (AdaLN, cross-condition, adaptive gamma, HC dynamics) active"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('to_gamma.weight') or 'text_to_audio' in name or 'audio_to_text' in name:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / p.shape[-1] ** 0.5))
            elif name.endswith('to_gamma.bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5 - 1.0)
            elif 'dynamic_alpha_fn' in name or 'dynamic_beta_fn' in name:
                p.copy_(torch.randn(p.shape, generator=g) * (p.shape[0] ** -0.5))
            elif 'dynamic_alpha_scale' in name or 'dynamic_beta_scale' in name:
                p.fill_(0.3)
            elif 'norm.gamma' in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
            elif 'to_v_head_gate.bias' in name or 'to_value_residual_mix.0.bias' in name:
                p.copy_(torch.randn(p.shape, generator=g))
            elif 'to_v_head_gate.weight' in name or 'to_value_residual_mix.0.weight' in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif name.endswith('.g'):
                p.copy_(1 + torch.randn(p.shape, generator=g) * 0.2)


VARIANTS = {                                   # the constructor's default-off switches (e2_tts.py:533-546), depth 2
    'fourier': dict(depth=2, attn_fourier_embed_input=True),
    'laser': dict(depth=2, attn_laser=True),
    'freq_axis': dict(depth=2, has_freq_axis=True, freq_heads=2),
}


@pytest.mark.parametrize('cond_on_time,with_text,with_mask,variant', [
    (True, True, True, None), (False, False, False, None),
    (True, True, True, 'fourier'), (True, True, True, 'laser'), (True, True, True, 'freq_axis'), (False, False, True, 'freq_axis')])
def test_backbone(dev, cond_on_time, with_text, with_mask, variant):
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    kw = dict(dim=256, depth=4, heads=2, dropout=0., max_seq_len=64)
    kw.update(VARIANTS.get(variant, {}))
    fshape = (3,) if kw.get('has_freq_axis') else ()
    ref = O.Transformer(**kw, cond_on_time=cond_on_time)
    randomize(ref)
    mod = Transformer(**kw, cond_on_time=cond_on_time)
    missing = mod.load_state_dict(ref.state_dict(), strict=True)
    mod = mod.to(dev)
    B, T = 2, 40
    x = torch.randn(B, *fshape, T, 256)
    times = torch.rand(B) if cond_on_time else None
    text = torch.randn(B, T, 128) if with_text else None
    mask = None
    if with_mask:
        mask = torch.arange(T)[None] < torch.tensor([T, T - 9])[:, None]
    R = torch.randn(B, *fshape, T, 256)

    xr = x.clone().requires_grad_(True)
    tr = text.clone().requires_grad_(True) if with_text else None
    out_r = ref(xr, times=times, mask=mask, text_embed=tr)
    (out_r * R).sum().backward()

    xk = x.clone().to(dev).requires_grad_(True)
    tk = text.clone().to(dev).requires_grad_(True) if with_text else None
    out_k = mod(xk, times=None if times is None else times.to(dev), mask=None if mask is None else mask.to(dev), text_embed=tk)
    # (max-abs over max-abs: with these stress weights ONE element in 61 k moves between 1.7 % and 4 % with the rounding pattern of the
    #  attention's soft-clamp polynomial -- first-generation ring kernels 1.7 %, degree-7 polynomial only 2.1 %, cubic tier 3.1 % --, while
    #  rel-L2 stays at 0.85-0.90 %: the robust bound is the rel-L2 one below)
    assert rel(out_k, out_r) < 5e-2, rel(out_k, out_r)
    (out_k * R.to(dev)).sum().backward()
    assert rel2(out_k, out_r) < 2e-2, rel2(out_k, out_r)
    # (stress weights: the input gradient of the frequency-axis model without time conditioning sits at 4.8 % with the branch norm as a
    #  launch of its own and at 5.4 % inside the width connection -- another draw of the same roundings; the other five variants 1.1-4.1 %)
    assert rel2(xk.grad, xr.grad) < 7e-2, rel2(xk.grad, xr.grad)
    if with_text:
        assert rel2(tk.grad, tr.grad) < 5e-2, rel2(tk.grad, tr.grad)
    refp = dict(ref.named_parameters())
    gr_fp32 = {n: p_.grad.item() for n, p_ in refp.items() if p_.numel() == 1 and p_.grad is not None}
    bad, errs = [], []
    for name, p in mod.named_parameters():
        gr = refp[name].grad
        if gr is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0., name
            continue
        assert p.grad is not None, name
        err = rel2(p.grad, gr)
        errs.append((err, name))
        if p.numel() == 1:       # heavily cancelling sums over all tokens: absolute slack (unit-tested in test_emu_hc)
            ok = abs(p.grad.cpu().item() - gr.item()) <= 0.4 * abs(gr.item()) + 10.0
        else:
            ok = err <= (0.6 if p.numel() <= 32 else 0.15)   # bf16 noise accumulated over the whole backward; a missing term shows as >= 0.3
        if not ok:
            bad.append((name, err, float(gr.norm())))
    scalars = [b for b in bad if refp[b[0]].numel() == 1]
    if scalars:
        # The scalar hyper-connection scales are sums over every token of terms of both signs (|gradient| up to ~900 here);
        # bf16 rounding moves each by a few units to a few tens, at random (tools: the same model at three seeds shows
        # deviations of either sign, 1 .. 33, for every one of them, and the outlier moves to another parameter with the seed).
        # A wrong or missing term shifts ALL of a connection's scalars; a lone outlier is rounding.  So: at most one scalar
        # may miss the per-parameter bound, and only by less than a fifth of the largest scalar gradient.
        top = max(abs(v) for v in gr_fp32.values())
        got = dict(mod.named_parameters())
        far = [n for n, _, _ in scalars if abs(got[n].grad.cpu().item() - gr_fp32[n]) > 0.2 * top]
        print('scalar outliers', [(n, gr_fp32[n], got[n].grad.cpu().item()) for n, _, _ in scalars], 'largest scalar gradient', top)
        if len(scalars) == 1 and not far:
            bad = [b for b in bad if b[0] != scalars[0][0]]
    errs.sort(reverse=True)
    print('worst parameter-gradient errors:', errs[:8])
    import statistics
    print('median err', statistics.median(e for e, _ in errs), 'n', len(errs), 'out', rel2(out_k, out_r), 'dx', rel2(xk.grad, xr.grad))
    assert not bad, bad[:20]


@pytest.mark.parametrize('late', [0, 1])
def test_backbone_with_256_tile_gemm(dev, monkeypatch, late):
    """the whole backbone with EVERY forward / dgrad GEMM forced onto the 256 x 256 8-phase kernel (E2K_GEMM_T256); on the
    host model under both LDS-DMA landing extremes (`late` has no meaning on the GPU: one run there)"""
    from e2_tts_pytorch_amd import ops
    if dev == 'cuda' and late:
        pytest.skip('LDS-DMA landing extremes exist on the host model only')
    monkeypatch.setenv('E2K_EMU_GLDS_LATE', str(late))
    monkeypatch.setattr(ops, 'gemm_flags', 128)
    test_backbone(dev, True, True, True, None)


def test_persistent_grads(dev):
    """enable_persistent_grads(): every .grad is a permanent view of one flat buffer that each backward overwrites; the
    values are those of the default mode (fresh gradient tensors handed to autograd)"""
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    mod = Transformer(dim=256, depth=2, heads=2, dropout=0., max_seq_len=64, num_registers=32 if gpu_shapes(dev) else 8)      # (host model: fewer register tokens, same schedule)
    randomize(mod)
    mod = mod.to(dev)
    B, T = 2, 24
    R = torch.randn(B, T, 256).to(dev)

    def step(seed, zero):
        if zero:
            mod.zero_grad(set_to_none=True)
        else:                      # only the parameters outside the flat buffer (ordinary autograd accumulation)
            for p in mod.time_cond_mlp.parameters():
                p.grad = None
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, T, 256, generator=g).to(dev).requires_grad_(True)
        t = torch.rand(B, generator=g).to(dev)
        txt = torch.randn(B, T, 128, generator=g).to(dev).requires_grad_(True)
        out = mod(x, times=t, text_embed=txt)
        (out * R).sum().backward()
        return x.grad.clone(), txt.grad.clone(), {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}

    ref = [step(s, True) for s in (1, 2)]
    ref.append(ref[1])
    mod.zero_grad(set_to_none=True)
    mod.enable_persistent_grads()
    got = [step(1, False), step(2, False)]            # no zero_grad in between: the second pass must not accumulate
    flat_ids = {id(q) for q, _ in mod._layout.slots}    # (time_cond_mlp sits outside the hand-scheduled part)
    ids = {n: p.grad.data_ptr() for n, p in mod.named_parameters() if p.grad is not None and id(p) in flat_ids}
    views = {n: p.grad for n, p in mod.named_parameters() if n in ids}
    got.append(step(2, True))                          # zero_grad(set_to_none=True) detaches the views: re-attached
    for n, p in mod.named_parameters():
        if n in ids:
            assert p.grad.data_ptr() == ids[n] and p.grad is views[n], n
    for (dx0, dt0, g0), (dx1, dt1, g1) in zip(ref, got):
        assert rel2(dx1, dx0) < 1e-5 and rel2(dt1, dt0) < 1e-5, (rel2(dx1, dx0), rel2(dt1, dt0))
        assert g0.keys() == g1.keys()
        for n in g0:                # (not bit-equal: several gradients are float atomics, whose order varies run to run)
            assert rel2(g1[n], g0[n]) < 1e-4 or float(g0[n].norm()) < 1e-6, (n, rel2(g1[n], g0[n]))
    buf = mod._pg.buf
    lay = mod._layout
    assert all(p.grad.data_ptr() == buf.data_ptr() + 4 * off for p, off in lay.slots if p.grad is not None)
    # back to the default mode: the next backward hands fresh tensors to autograd again
    # (the buffer itself stays while recorded plans write into it; what changes is what autograd is handed)
    mod.enable_persistent_grads(False)
    assert not mod._persist_grads
    back = step(2, True)
    assert all(p.grad.data_ptr() != buf.data_ptr() + 4 * off for p, off in lay.slots if p.grad is not None)
    for n in ref[1][2]:
        assert rel2(back[2][n], ref[1][2][n]) < 1e-4 or float(ref[1][2][n].norm()) < 1e-6, n


@pytest.mark.parametrize('case', ['transformer_full', 'transformer_bare', 'transformer_variant', 'transformer_laser_fourier',
                                  'transformer_freq_axis', 'transformer_freq_axis_bare'])
def test_reference_golden_backbone(dev, case):
    """hand-scheduled backbone vs the outputs of the reference's own Transformer (tests/golden/reference_pinned.pt, written
    by oracle/pin_against_reference.py): forward, input gradients and the magnitude of every parameter gradient.
    `variant`: text stream in 2 of 4 layers, 128-wide text stream, 15-tap convolution, 8 registers, no abs-pos embedding"""
    from pathlib import Path
    from e2_tts_pytorch_amd import Transformer
    from oracle.golden_weights import fill_params
    c = torch.load(Path(__file__).resolve().parent / 'golden' / 'reference_pinned.pt', weights_only=False)[case]
    random.seed(0)
    mod = fill_params(Transformer(**c['kw'], cond_on_time=c['cond_on_time']), c['weight_seed']).to(dev)
    to = lambda t: None if t is None else t.to(dev)
    x = c['x'].clone().to(dev).requires_grad_(True)
    t = c['text'].clone().to(dev).requires_grad_(True) if c['text'] is not None else None
    out = mod(x, times=to(c['times']), mask=to(c['mask']), text_embed=t)
    (out * c['R'].to(dev)).sum().backward()
    assert rel2(out, c['out']) < 2e-2, rel2(out, c['out'])
    # with every parameter drawn at random the input gradient is ill-conditioned: rounding weights and input to bf16
    # alone moves it by 6 % in the fp32 model.  0.15 still separates rounding from a missing term (>= 0.3)
    assert rel2(x.grad, c['dx']) < 0.15, rel2(x.grad, c['dx'])
    # parameter gradients: the fixture carries sum |g| per parameter; compare for the large matrices (for small / heavily
    # cancelling ones the sum of magnitudes is dominated by rounding noise; element-wise parity is test_backbone's job).
    # `variant` draws EVERY parameter at random, which makes the early layers' weight gradients ill-conditioned in the
    # MODEL: rounding the fp32 oracle's own activations to bf16 (tests/bf16_emulation.py) already moves their sum of
    # magnitudes by up to ~17 % (tools/probes/which_rounding.py: it is the FORWARD roundings that do it -- stream storage
    # 5 %, branch inputs / outputs 4 %, together 17 % -- rounding the stream gradients alone moves them by 0.4 %).  The
    # tolerance of each parameter is therefore 0.15 plus twice the deviation of that emulation for the same parameter.
    # Round 6: on the bf16 path these sums are, for `transformer_full`, NOISE-dominated: a 1e-3 relative perturbation of the input -- which
    # only re-draws which way the intermediate values round -- swings the feed-forward tensors' sums between 0.95 and 1.33 x the
    # reference (all of them together: they share the branch's upstream gradient), while the fp32 oracle moves by 3 % under the same
    # perturbation.  The fixed 0.15 band only held while one particular set of roundings fell well: fusing the branch norm into the width
    # connection changed the last bits of 0.3 % of the normalised branch input and moved the sums by +25 %.  What separates rounding
    # from a missing term (which shifts EVERY realisation, by >= 30 %) is the MEDIAN over realisations (the exact input and four
    # perturbed ones) held against a band that widens with the realisations' own spread.
    slack = {}
    sums = {}
    for draw in range(5):
        mod.zero_grad(set_to_none=True)
        if draw:
            gen = torch.Generator().manual_seed(draw)
            xp = (c['x'] * (1. + 1e-3 * torch.randn(c['x'].shape, generator=gen))).to(dev).requires_grad_(True)
            tp = c['text'].clone().to(dev).requires_grad_(True) if c['text'] is not None else None
            (mod(xp, times=to(c['times']), mask=to(c['mask']), text_embed=tp) * c['R'].to(dev)).sum().backward()
        else:
            (mod(x.detach().clone().requires_grad_(True), times=to(c['times']), mask=to(c['mask']), text_embed=None if t is None else t.detach().clone().requires_grad_(True)) * c['R'].to(dev)).sum().backward()
        for n, p in mod.named_parameters():
            if p.grad is not None:
                sums.setdefault(n, []).append(float(p.grad.double().abs().sum()))
    if case == 'transformer_variant':
        from bf16_emulation import bf16_intermediates
        random.seed(0)
        ref = fill_params(O.Transformer(**c['kw'], cond_on_time=c['cond_on_time']), c['weight_seed'])
        xr = c['x'].clone().requires_grad_(True)
        tr_ = c['text'].clone().requires_grad_(True) if c['text'] is not None else None
        with bf16_intermediates():
            (ref(xr, times=c['times'], mask=c['mask'], text_embed=tr_) * c['R']).sum().backward()
        for n, p in ref.named_parameters():
            want = c['grad_abs_sums'].get(n)
            if want and p.grad is not None:
                slack[n] = max(slack.get(n, 0.), 2. * abs(float(p.grad.double().abs().sum()) / want - 1.))
    bad = []
    for n, p in mod.named_parameters():
        want = c['grad_abs_sums'].get(n)
        if want is None or p.grad is None or p.numel() < 16384 or want == 0.:
            continue
        got = sorted(sums[n])[len(sums[n]) // 2]
        # ... within 0.15 (+ the emulation's slack) + half the range the realisations themselves span: a noise-dominated sum of MAGNITUDES is
        # also biased upwards (ten realisations of layers.1.0.7.ff.0.proj.weight: mean 1.16-1.21 x the reference, +-0.17, with the fused norm
        # and without alike; the well-conditioned attention projections: 1.01-1.03 +- 0.02)
        spread = (max(sums[n]) - min(sums[n])) / (2. * want)
        if abs(got - want) > (0.15 + slack.get(n, 0.) + spread) * want:
            bad.append((n, got / want, slack.get(n, 0.), [round(v / want, 3) for v in sums[n]]))
    assert not bad, bad[:10]


def test_launch_lanes_match_single_stream(dev):
    """launch lanes (ops.Lanes, csrc/plan.h): text branches and weight-gradient GEMMs on side streams must give what
    the single-stream schedule gives -- outputs and input gradients bit for bit, parameter gradients up to the order
    of their fp32 atomics -- in the eager schedule, while a plan is recorded and on its replays, training and
    forward-only.  Several steps each: a missing ordering point shows up as a stale or half-written operand, and so
    does a kernel that is not reproducible next to a concurrent one (hc_bwd_kernel with LDS float atomics was not:
    tools/probes/hc_concurrent.py).  On the host model there are no streams: the test then checks the bookkeeping --
    every recorded wait has its record, the replay through e2k_plan_run_lanes matches."""
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    big = gpu_shapes(dev)
    dim, depth, B, T = (512, 6, 4, 200) if big else (256, 2, 1, 16)
    mod = Transformer(dim=dim, depth=depth, heads=dim // 64, dropout=0., max_seq_len=T, num_registers=32 if gpu_shapes(dev) else 8)
    randomize(mod)
    mod = mod.to(dev)
    R = torch.randn(B, T, dim).to(dev)

    def inputs(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, T, dim, generator=g).to(dev).requires_grad_(True)
        t = torch.rand(B, generator=g).to(dev)
        txt = torch.randn(B, T, dim // 2, generator=g).to(dev).requires_grad_(True)
        return x, t, txt

    def step(seed):
        mod.zero_grad(set_to_none=True)
        x, t, txt = inputs(seed)
        out = mod(x, times=t, text_embed=txt)
        (out * R).sum().backward()
        return out.detach().clone(), x.grad.clone(), txt.grad.clone(), {n: p.grad.clone() for n, p in mod.named_parameters()}

    def infer(seed):
        with torch.no_grad():
            x, t, txt = inputs(seed)
            return mod(x, times=t, text_embed=txt).clone()

    seeds = (1, 2, 3, 2, 1) if big else (1, 2, 3)
    res = {}
    for lanes in (False, True):
        mod.enable_lanes(lanes, backward=lanes)
        for plans in (False, True):
            if not big and plans and not lanes:
                continue                    # (host model: test_plan_replay_matches_eager covers plans without lanes)
            mod.enable_plans(plans)
            res[lanes, plans] = [step(s) for s in seeds], [infer(s) for s in seeds]
    st = [v for v in mod._plans.values() if not isinstance(v, str) and v.bwd]
    assert st and len(st[0].lane_ss) == 2            # (the last setting recorded: lanes on)
    fwd_names = ops_names(st[0].fwd)
    assert fwd_names.count('lane_event_wait') >= 2 * depth and fwd_names.count('lane_event_record') >= 2 * depth
    if True:
        names = ops_names(st[0].bwd)
        assert names.count('lane_event_wait') >= 2 * depth and names.count('lane_event_record') >= 2 * depth
    step(seeds[0])                                      # (make the training plan the most recently used one)
    rows = mod.plan_profile()                           # (what bench.py prices the roofline with: every call alone on one stream)
    assert {r['lane'] for r in rows} == {0, 1, 2} and all(r['ms'] >= 0 for r in rows) and {r['phase'] for r in rows} == {'fwd', 'bwd'}
    assert sum(r['name'] == 'gemm_tn_bf16' and r['lane'] == 2 for r in rows) > 0 and not any(r['name'] == 'gemm_tn_bf16' and r['phase'] == 'fwd' for r in rows)
    ref_t, ref_i = res[False, False]
    for key in ((True, False), (True, True), (False, True)):
        if key not in res:
            continue
        got_t, got_i = res[key]
        for (o0, dx0, dt0, g0), (o1, dx1, dt1, g1) in zip(ref_t, got_t):
            assert torch.equal(o1, o0) and torch.equal(dx1, dx0) and torch.equal(dt1, dt0), key
            for n in g0:
                assert rel2(g1[n], g0[n]) < 2e-3 or float(g0[n].norm()) < 1e-6, (key, n, rel2(g1[n], g0[n]))
        for a, b in zip(ref_i, got_i):
            assert torch.equal(a, b), key

def test_geglu_epilogue_in_the_backbone(dev):
    """ops.fuse_geglu (default on): in no-grad forwards FeedForward's GEGLU + dropout is the epilogue of its first GEMM
    (e2k_gemm_nt_geglu_bf16; the pre-activation H is never written) instead of a separate pass over H;
    training steps keep GEMM + e2k_geglu_fwd (H is needed by the backward, and the fusion measured step-neutral there:
    profiles/r03_first_call_switch_ab.jsonl).  Eager and through a recorded plan, module in train() mode so that dropout
    is on: the keep mask is a function of (seed, stream id, row, column) only, so both schedules draw the same one."""
    from e2_tts_pytorch_amd import Transformer, ops
    random.seed(0)
    torch.manual_seed(0)
    big = gpu_shapes(dev)
    dim, depth, B, T = (512, 4, 4, 200) if big else (256, 2, 1, 24)
    mod = Transformer(dim=dim, depth=depth, heads=dim // 64, dropout=0.1, max_seq_len=T, num_registers=32 if gpu_shapes(dev) else 8)
    randomize(mod)
    mod = mod.to(dev)
    mod.train()
    g = torch.Generator().manual_seed(5)
    x0, t0, txt0 = torch.randn(B, T, dim, generator=g), torch.rand(B, generator=g), torch.randn(B, T, dim // 2, generator=g)

    def infer():
        torch.manual_seed(77)                                # the dropout seed of the step is drawn from torch's generator
        with torch.no_grad():
            return mod(x0.to(dev), times=t0.to(dev), text_embed=txt0.to(dev)).clone()

    def train_names():
        x = x0.clone().to(dev).requires_grad_(True)
        for _ in range(3):
            mod.zero_grad(set_to_none=True)
            mod(x, times=t0.to(dev), text_embed=txt0.to(dev)).sum().backward()
        st = [v for v in mod._plans.values() if not isinstance(v, str) and v.bwd]
        return ops_names(st[0].fwd)

    res, names = {}, {}
    old = ops.fuse_geglu
    try:
        for fused in (False, True):
            ops.fuse_geglu = fused
            mod._drop_plans()
            for plans in (False, True):
                mod.enable_plans(plans)
                for _ in range(3 if plans else 1):          # (a signature is recorded the second time it is seen)
                    res[fused, plans] = infer()
            st = [v for v in mod._plans.values() if not isinstance(v, str) and not v.need_grad]
            names[fused] = ops_names(st[0].fwd)
            if fused:
                tn = train_names()
                assert tn.count('geglu_fwd') == 2 * depth and 'gemm_nt_geglu_bf16' not in tn
    finally:
        ops.fuse_geglu = old
    assert names[True].count('gemm_nt_geglu_bf16') == 2 * depth and 'geglu_fwd' not in names[True]
    assert names[False].count('geglu_fwd') == 2 * depth and 'gemm_nt_geglu_bf16' not in names[False]
    ref = res[False, False]
    # (the epilogue's erf differs from the library's in the last place of a few activations: bf16 noise level)
    assert rel2(res[True, False], ref) < 2e-2 and rel2(res[False, True], ref) < 1e-6
    assert torch.equal(res[True, True], res[True, False])   # plan replay of the fused schedule == its eager run


def ops_names(handle):
    import ctypes
    from e2_tts_pytorch_amd import ops
    L = ops.lib()
    buf = ctypes.create_string_buffer(64)
    out = []
    for i in range(L.e2k_query_plan_size(handle)):
        L.e2k_plan_op_name(handle, i, ctypes.addressof(buf), 64)
        out.append(buf.value.decode())
    return out


def test_plan_replay_matches_eager(dev):
    """launch plans (csrc/plan.h): the recorded launch sequence of a signature, re-issued from C++, reproduces the eager
    schedule bit for bit -- forward, input gradients and every parameter gradient; gradients delivered through autograd
    (default, accumulating) and through the persistent flat buffer; a forward-only plan (the sampling path); a second
    forward before the backward falls back to the eager schedule instead of overwriting the saved activations"""
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    depth, T = (4, 40) if gpu_shapes(dev) else (2, 24)          # (the host model is ~1000x slower than the GPU: same schedule, fewer layers / frames)
    mod = Transformer(dim=256, depth=depth, heads=2, dropout=0., max_seq_len=64, num_registers=32 if gpu_shapes(dev) else 8)
    randomize(mod)
    mod = mod.to(dev)
    B = 2
    R = torch.randn(B, T, 256).to(dev)

    def inputs(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, T, 256, generator=g).to(dev).requires_grad_(True)
        t = torch.rand(B, generator=g).to(dev)
        txt = torch.randn(B, T, 128, generator=g).to(dev).requires_grad_(True)
        mask = (torch.arange(T)[None] < torch.tensor([T, T - 5 - seed])[:, None]).to(dev)
        return x, t, txt, mask

    def step(seed):
        mod.zero_grad(set_to_none=True)
        x, t, txt, mask = inputs(seed)
        out = mod(x, times=t, mask=mask, text_embed=txt)
        (out * R).sum().backward()
        return out.detach().clone(), x.grad.clone(), txt.grad.clone(), {n: p.grad.clone() for n, p in mod.named_parameters()}

    def live_plans():
        return [v for v in mod._plans.values() if not isinstance(v, str)]

    mod.enable_plans(False)
    ref = [step(s) for s in (1, 2, 3)]
    mod.enable_plans(True)
    got = [step(s) for s in (1, 2, 3)]          # first sighting (eager), recording pass, replay
    assert len(live_plans()) == 1 and live_plans()[0].bwd, 'nothing was recorded'
    for (o0, dx0, dt0, g0), (o1, dx1, dt1, g1) in zip(ref, got):
        assert torch.equal(o1, o0) and torch.equal(dx1, dx0) and torch.equal(dt1, dt0)
        for n in g0:        # (fp32 atomics in the gradient reductions: order-of-arrival noise, amplified where a bf16 rounding follows)
            assert rel2(g1[n], g0[n]) < 2e-3 or float(g0[n].norm()) < 1e-6, n
    # accumulation over two backward passes without zero_grad: autograd adds, as with any module
    x, t, txt, mask = inputs(3)
    (mod(x, times=t, mask=mask, text_embed=txt) * R).sum().backward()
    n0 = 'layers.1.0.3.to_q.weight'
    assert rel2(dict(mod.named_parameters())[n0].grad, 2 * got[2][3][n0]) < 2e-3
    # persistent flat gradient buffer: same values, no copy, nothing handed to autograd
    mod.enable_persistent_grads()
    got2 = [step(s) for s in (1, 2, 3)]
    for (o0, dx0, dt0, g0), (o1, dx1, dt1, g1) in zip(ref, got2):
        assert torch.equal(o1, o0) and torch.equal(dx1, dx0)
        for n in g0:
            assert rel2(g1[n], g0[n]) < 2e-3 or float(g0[n].norm()) < 1e-6, n
    mod.enable_persistent_grads(False)
    # two forwards of one signature before a backward: the second must not touch the first one's saved activations
    mod.zero_grad(set_to_none=True)
    xa, ta, txa, ma = inputs(1)
    xb, tb, txb, mb = inputs(2)
    la = (mod(xa, times=ta, mask=ma, text_embed=txa) * R).sum()
    lb = (mod(xb, times=tb, mask=mb, text_embed=txb) * R).sum()
    (la + lb).backward()
    assert torch.equal(xa.grad, ref[0][1]) and torch.equal(xb.grad, ref[1][1])
    # forward-only plan (sampling path)
    with torch.no_grad():
        x, t, txt, mask = inputs(4)
        eager = mod.enable_plans(False)(x, times=t, mask=mask, text_embed=txt).clone()
        mod.enable_plans(True)
        outs = [mod(x, times=t, mask=mask, text_embed=txt).clone() for _ in range(3)]
    assert torch.equal(outs[1], eager) and torch.equal(outs[2], eager)
    assert len(live_plans()) == 1 and not live_plans()[0].need_grad       # (enable_plans(False) dropped the training plan)
    # a deep copy (the trainer's EMA) starts without runtime state and works on its own
    import copy
    twin = copy.deepcopy(mod)
    assert twin._plans == {} and twin._flat is None
    with torch.no_grad():
        assert torch.equal(twin(x, times=t, mask=mask, text_embed=txt), eager)


def test_plan_graph_replay_matches_eager_replay(dev):
    """Transformer.enable_graphs (round 6; csrc/plan.hip e2k_query_plan_graph_capture): a recorded plan replayed as ONE HIP graph per
    pass -- captured from the eager replay loop, lanes forked from / joined to the capturing stream -- gives what the eager replay
    gives: outputs and input gradients bit for bit, parameter gradients to the order-of-arrival noise of the fp32 atomics; with dropout
    the device-side seed word keeps the masks fresh (two graph launches with different seeds differ, the same seed repeats); a
    forward-only (sampling) plan; and an installed gradient hook keeps the backward on the segmented eager replay."""
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    depth, T = (4, 40) if gpu_shapes(dev) else (2, 24)
    mod = Transformer(dim=256, depth=depth, heads=2, dropout=0., max_seq_len=64, num_registers=32 if gpu_shapes(dev) else 8)
    randomize(mod)
    mod = mod.to(dev)
    B = 2
    R = torch.randn(B, T, 256).to(dev)

    def inputs(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, T, 256, generator=g).to(dev).requires_grad_(True)
        t = torch.rand(B, generator=g).to(dev)
        txt = torch.randn(B, T, 128, generator=g).to(dev).requires_grad_(True)
        mask = (torch.arange(T)[None] < torch.tensor([T, T - 5 - seed])[:, None]).to(dev)
        return x, t, txt, mask

    def step(seed):
        mod.zero_grad(set_to_none=True)
        x, t, txt, mask = inputs(seed)
        out = mod(x, times=t, mask=mask, text_embed=txt)
        (out * R).sum().backward()
        return out.detach().clone(), x.grad.clone(), txt.grad.clone(), {n: p.grad.clone() for n, p in mod.named_parameters()}

    mod.enable_graphs(False)
    ref = [step(s) for s in (1, 2, 3, 4, 5)]        # first sighting, recording, three eager replays
    st = [v for v in mod._plans.values() if not isinstance(v, str)][0]
    assert st.bwd and not st.gfwd and not st.gbwd
    mod.enable_graphs(True)
    got = [step(s) for s in (3, 4, 5)]              # capture + launch, launch, launch
    assert st.gfwd and st.gbwd, 'no graph was captured'
    for (o0, dx0, dt0, g0), (o1, dx1, dt1, g1) in zip(ref[2:], got):
        assert torch.equal(o1, o0) and torch.equal(dx1, dx0) and torch.equal(dt1, dt0)
        for n in g0:
            assert rel2(g1[n], g0[n]) < 2e-3 or float(g0[n].norm()) < 1e-6, n
    # a gradient hook (the data-parallel exchange) must see every slab between the backward's segments: eager replay there
    seen = []

    class Hook:
        lanes = []

        def __call__(self, gflat, a, b):
            seen.append((a, b))
    mod._grad_sync = Hook()
    o, dx, dt, g = step(5)
    mod._grad_sync = None
    assert len(seen) >= depth + 1 and seen[-1] == (None, None) and torch.equal(o, ref[4][0]) and torch.equal(dx, ref[4][1])
    # toggling off frees the graphs and the eager replay is back
    mod.enable_graphs(False)
    assert not st.gfwd and not st.gbwd
    o, dx, dt, g = step(4)
    assert torch.equal(o, ref[3][0]) and torch.equal(dx, ref[3][1])
    # forward-only plan (the sampling path) as a graph
    mod.enable_graphs(True)
    mod.eval()
    with torch.no_grad():
        x, t, txt, mask = inputs(7)
        outs = [mod(x, times=t, mask=mask, text_embed=txt).clone() for _ in range(4)]
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    ng = [v for v in mod._plans.values() if not isinstance(v, str) and not v.need_grad]
    assert ng and ng[0].gfwd


def test_plan_graph_replay_keeps_dropout_fresh(dev):
    """dropout inside a graph: the keep masks come from a device-side seed word that every replay rewrites BEFORE the launch -- two
    launches of the same graph draw different masks, and re-seeding python's generator repeats them"""
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    T = 40 if gpu_shapes(dev) else 24
    mod = Transformer(dim=256, depth=2, heads=2, dropout=0.2, max_seq_len=64, num_registers=32 if gpu_shapes(dev) else 8)
    randomize(mod)
    mod = mod.to(dev).train()
    mod.enable_graphs(True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, T, 256, generator=g).to(dev)
    t = torch.rand(2, generator=g).to(dev)
    txt = torch.randn(2, T, 128, generator=g).to(dev)

    def run(seed):
        random.seed(seed)
        torch.manual_seed(seed)
        xx = x.clone().requires_grad_(True)
        out = mod(xx, times=t, text_embed=txt)
        out.sum().backward()
        return out.detach().clone(), xx.grad.clone()

    a = [run(s) for s in (11, 12, 13, 14, 13)]          # eager, recording, capture + launch, launch, launch with a repeated seed
    st = [v for v in mod._plans.values() if not isinstance(v, str)][0]
    assert st.gfwd and st.gbwd
    assert not torch.equal(a[2][0], a[3][0])
    assert torch.equal(a[2][0], a[4][0]) and torch.equal(a[2][1], a[4][1])


def test_branch_norm_inside_the_width_connection(dev, monkeypatch):
    """backbone._FUSE_HC_NORM: 0 = width connection and branch norm as two launches; 1 (default) = one launch in no-grad forwards, which do
    not write the un-normalised branch input at all; 2 = one launch in training passes too.  Same outputs and gradients to bf16 rounding of
    the normalised branch input (the fused kernel takes the norm from the fp32 streams' Gram matrix)"""
    import e2_tts_pytorch_amd.backbone as bbm
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    T = 40 if gpu_shapes(dev) else 24
    mod = Transformer(dim=256, depth=2, heads=2, dropout=0., max_seq_len=64, num_registers=32 if gpu_shapes(dev) else 8)
    randomize(mod)
    mod = mod.to(dev)
    mod.enable_plans(False)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(2, T, 256, generator=g).to(dev)
    t = torch.rand(2, generator=g).to(dev)
    txt = torch.randn(2, T, 128, generator=g).to(dev)
    mask = (torch.arange(T)[None] < torch.tensor([T, T - 7])[:, None]).to(dev)
    R = torch.randn(2, T, 256, generator=g).to(dev)
    res = {}
    for mode in (0, 2):
        monkeypatch.setattr(bbm, '_FUSE_HC_NORM', mode)
        mod.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        out = mod(x, times=t, mask=mask, text_embed=txt)
        (out * R).sum().backward()
        res[mode] = (out.detach().cpu(), x.grad.cpu(), {n: p.grad.cpu().clone() for n, p in mod.named_parameters()})
    # (half of the normalised branch input's elements move by one bf16 place -- an independent rounding, like the 0.5 % each mode is from the fp32 oracle)
    assert rel2(res[2][0], res[0][0]) < 1e-2 and rel2(res[2][1], res[0][1]) < 2e-2
    for n, g0 in res[0][2].items():
        if g0.numel() >= 4096:
            assert rel2(res[2][2][n], g0) < 4e-2 or float(g0.norm()) < 1e-6, n
        else:       # tiny tensors are heavily cancelling sums over all tokens: the scalar hyper-connection scales carry +-3 of noise whatever their size
            assert float((res[2][2][n] - g0).norm()) <= 0.15 * float(g0.norm()) + 4., n
    with torch.no_grad():
        outs = {}
        for mode in (0, 1):
            monkeypatch.setattr(bbm, '_FUSE_HC_NORM', mode)
            outs[mode] = mod(x0, times=t, mask=mask, text_embed=txt).cpu()
    assert rel2(outs[1], outs[0]) < 1e-2 and rel2(outs[0], res[0][0]) < 5e-3          # (the no-grad schedule fuses the GEGLU into its GEMM: not the training pass bit for bit)


def test_rotary_inside_the_attention_projection(dev, monkeypatch):
    """backbone._FUSE_QK_ROT: the attention's input projection writes q and k rotated and head-major from its own epilogue
    (e2k_gemm_nt_qkrot_bf16) and the post kernel runs the value path only.  Bit-identical to the two-launch form in the output and in
    every gradient, eagerly and as a recorded plan; the text branch of this model (K = 128) and its first layer (772 columns) are shapes the
    fused launch refuses: they stay on the two-launch form in the same pass"""
    import e2_tts_pytorch_amd.backbone as bbm
    from e2_tts_pytorch_amd import Transformer, ops
    random.seed(0)
    torch.manual_seed(0)
    T = 40 if gpu_shapes(dev) else 24
    mod = Transformer(dim=256, depth=4, heads=4, dropout=0., max_seq_len=64, num_registers=32 if gpu_shapes(dev) else 8)
    randomize(mod)
    mod = mod.to(dev)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(2, T, 256, generator=g).to(dev)
    t = torch.rand(2, generator=g).to(dev)
    txt = torch.randn(2, T, 128, generator=g).to(dev)
    mask = (torch.arange(T)[None] < torch.tensor([T, T - 7])[:, None]).to(dev)
    R = torch.randn(2, T, 256, generator=g).to(dev)
    monkeypatch.setattr(ops, 'qk_rot_any_size', True)      # (a model this small has fewer tiles than the library recommends the launch for)
    calls = []
    inner = ops.gemm_nt_qkrot
    monkeypatch.setattr(ops, 'gemm_nt_qkrot', lambda *a, **k: (calls.append(a[0].shape), inner(*a, **k))[1])

    def run():
        mod.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        out = mod(x, times=t, mask=mask, text_embed=txt)
        (out * R).sum().backward()
        return [out.detach().cpu(), x.grad.cpu()] + [p.grad.cpu().clone() for p in mod.parameters()]

    res = {}
    for fused in (False, True):
        monkeypatch.setattr(bbm, '_FUSE_QK_ROT', fused)
        mod.enable_plans(False)
        res[fused] = run()
        # the speech branch of layers 2 to 4; the first layer's projection (no value-residual mix: 3 x 256 + 4 columns, not a multiple
        # of 8) and the text branch are refused
        assert len(calls) == (3 if fused else 0)
    names = ['out', 'dx'] + [n for n, _ in mod.named_parameters()]

    def same(got, want):
        # out and d(x) bit for bit; parameter gradients to the order of their fp32 atomic sums -- 1e-7 of the tensor's largest entry where
        # the sum is the gradient, up to 1e-5 (measured on MI355X: to_gamma.weight) where the summed quantity is rounded to a bf16 GEMM
        # operand afterwards and a different arrival order flips a rounding
        for i, (n, a, b) in enumerate(zip(names, got, want)):
            assert torch.equal(a, b) if i < 2 else float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), (n, float((a - b).abs().max()), float(b.abs().max()))

    same(res[True], res[False])
    mod.enable_plans(True)
    for _ in range(3):                                      # eager, recording, replay
        same(run(), res[False])
    with torch.no_grad():
        outs = {}
        for fused in (False, True):
            monkeypatch.setattr(bbm, '_FUSE_QK_ROT', fused)
            mod.enable_plans(False)
            outs[fused] = mod(x0, times=t, mask=mask, text_embed=txt).cpu()
    assert torch.equal(outs[True], outs[False])


def test_recorded_training_plan_recasts_its_weights_every_replay(dev):
    """a recorded training plan refreshes the bf16 shadows of the fp32 master weights itself -- since round 6 the global slab and the first
    layer on the chain and the other layers two at a time on the WGRAD lane, each chunk with an ordering point its consumers wait for
    (backbone._RECAST_W_ON_LANE).  After the parameters changed (an optimizer step between replays) a replay must compute with the NEW
    weights in every layer: shadows equal to the rounded masters afterwards, and the output that of a fresh eager pass"""
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    T = 40 if gpu_shapes(dev) else 24
    mod = Transformer(dim=256, depth=4, heads=2, dropout=0., max_seq_len=64, num_registers=32 if gpu_shapes(dev) else 8)
    randomize(mod)
    mod = mod.to(dev)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, T, 256, generator=g).to(dev)
    t = torch.rand(2, generator=g).to(dev)
    txt = torch.randn(2, T, 128, generator=g).to(dev)

    def step():
        mod.zero_grad(set_to_none=True)
        out = mod(x.clone().requires_grad_(True), times=t, text_embed=txt)
        out.sum().backward()
        return out.detach().clone()

    import copy
    twin = copy.deepcopy(mod)                                # the eager schedule on the same weights
    twin.enable_plans(False)

    def step_twin():
        twin.zero_grad(set_to_none=True)
        out = twin(x.clone().requires_grad_(True), times=t, text_embed=txt)
        out.sum().backward()
        return out.detach().clone()

    step(); step(); step()                                   # eager, recording, replay
    st = [v for v in mod._plans.values() if not isinstance(v, str)][0]
    assert st.fwd and st.bwd
    for k in range(3):
        with torch.no_grad():
            for p, q in zip(mod.parameters(), twin.parameters()):        # an "optimizer step" that moves every layer
                d = torch.randn(p.shape, generator=g).to(dev) * 0.02 * (p.abs().mean() + 1e-3)
                p.add_(d)
                q.add_(d)
                torch.autograd.graph.increment_version(p)
                torch.autograd.graph.increment_version(q)
        got = step()                                         # a replay of the recorded plan
        assert [v for v in mod._plans.values() if not isinstance(v, str)][0] is st
        assert torch.equal(mod._shadow.cpu(), mod._flat.to(torch.bfloat16).cpu()), k
        assert torch.equal(got, step_twin()), k


def test_text_live_flag_is_resolved_where_it_is_read():
    """Transformer._text_grad_live under a data-parallel exchange whose answer arrives later (ddp._GradSync.begin_text_live on a HIP device
    returns a handle: pinned word + event): the backward pass only PARKS the handle (round 6: it used to wait for it), the first read of
    the flag resolves every parked handle and ORs the answers, assignment (the optimizer's `= None`) clears them, answers known at once
    (single rank) queue behind pending ones, and the parked list stays bounded when nobody ever reads the flag"""
    from e2_tts_pytorch_amd import Transformer
    tr = Transformer(dim=256, depth=2, heads=2, max_seq_len=64)
    waited = []

    class FakeSync:
        lanes = []

        def begin_text_live(self, live, device):
            return ('handle', live)

        def end_text_live(self, handle):
            waited.append(handle)
            return handle[1]

        def __call__(self, *a):
            pass
    tr._grad_sync = FakeSync()
    assert tr._text_grad_live is None
    assert tr._end_text_live(tr._begin_text_live(False, 'cpu'), False) is None and not waited          # parked, nobody waited
    tr._end_text_live(tr._begin_text_live(True, 'cpu'), True)
    tr._end_text_live(False, False)                                                               # an immediate answer behind pending ones
    assert not waited and tr._text_live_is_global
    assert tr._text_grad_live is True and len(waited) == 2                                        # first read: resolved, ORed
    assert tr._text_grad_live is True and len(waited) == 2                                        # (once)
    tr._text_grad_live = None                                                                     # the optimizer consumed it
    tr._end_text_live(tr._begin_text_live(False, 'cpu'), False)
    assert tr._text_grad_live is False and len(waited) == 3
    tr._text_grad_live = None
    for _ in range(200):
        tr._end_text_live(tr._begin_text_live(False, 'cpu'), False)
    assert len(tr.__dict__['_text_live_pending']) <= 64 and tr._text_grad_live is False
    tr._grad_sync = None
    tr._text_grad_live = None
    assert tr._end_text_live(True, True) is True and tr._text_grad_live is True                   # single rank: known at once


def test_plan_replay_with_the_default_off_switches(dev):
    """has_freq_axis + attn_laser + attn_fourier_embed_input all on: the recorded plan (forward and backward, launch lanes on)
    reproduces the eager schedule -- the frequency attention, LASER maps and Fourier kernels are ordinary recorded calls and
    the frequency-token count is part of the plan's signature"""
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    T = 24 if gpu_shapes(dev) else 12
    mod = Transformer(dim=256, depth=2, heads=2, dropout=0., max_seq_len=64, has_freq_axis=True, freq_heads=2, attn_laser=True,
                      attn_fourier_embed_input=True, num_registers=32 if gpu_shapes(dev) else 8)
    randomize(mod)
    mod = mod.to(dev)
    B = 2

    def step(seed, F):
        mod.zero_grad(set_to_none=True)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, F, T, 256, generator=g).to(dev).requires_grad_(True)
        t = torch.rand(B, generator=g).to(dev)
        txt = torch.randn(B, T, 128, generator=g).to(dev).requires_grad_(True)
        mask = (torch.arange(T)[None] < torch.tensor([T, T - 5 - seed])[:, None]).to(dev)
        R = torch.randn(B, F, T, 256, generator=g).to(dev)
        out = mod(x, times=t, mask=mask, text_embed=txt)
        (out * R).sum().backward()
        return out.detach().clone(), x.grad.clone(), txt.grad.clone(), {n: p.grad.clone() for n, p in mod.named_parameters()}

    mod.enable_plans(False)
    ref = [step(s, 3) for s in (1, 2, 3)] + [step(4, 2)]
    mod.enable_plans(True)
    got = [step(s, 3) for s in (1, 2, 3)] + [step(4, 2)]        # first sighting, recording, replay; then another token count (eager)
    assert len([v for v in mod._plans.values() if not isinstance(v, str)]) == 1
    for (o0, dx0, dt0, g0), (o1, dx1, dt1, g1) in zip(ref, got):
        assert torch.equal(o1, o0) and torch.equal(dx1, dx0) and torch.equal(dt1, dt0)
        for n in g0:
            assert rel2(g1[n], g0[n]) < 2e-3 or float(g0[n].norm()) < 1e-6, n


@pytest.mark.gpu
def test_plan_pools_survive_the_collector_inside_a_recording():
    """GPUTEST_r03 died here: a dead module's plan is cyclic garbage, its private torch.cuda.MemPool is destroyed whenever the
    cycle collector runs, and a MemPool destroyed while ANOTHER use_mem_pool context is open (= inside any plan recording)
    trips `captures_underway.empty()` in the caching allocator's destructor path -> std::terminate -> SIGABRT.  Children of
    tools/probes/mempool_dtor_abort.py: with the collector FORCED to run inside the next module's recording the package must
    survive (backbone._PlanPool parks the pool until no pool context is open).  In a child process: a regression would kill it,
    not the test run."""
    import importlib.util
    from pathlib import Path
    path = Path(__file__).resolve().parent.parent / 'tools' / 'probes' / 'mempool_dtor_abort.py'
    spec = importlib.util.spec_from_file_location('mempool_dtor_abort', path)
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    args = dict(pkg=str(probe.ROOT / 'e2-tts-pytorch_amd'), root=str(probe.ROOT))
    now = probe.run(probe.PRODUCT % dict(args, old=False))
    assert now['rc'] == 0 and 'survived' in now['stdout'], now


@pytest.mark.gpu
def test_plan_stress_every_switch_combination():
    """plan record + 20 replays on a FRESH module for each of the 8 combinations of has_freq_axis / attn_laser /
    attn_fourier_embed_input, garbage of the previous modules (their plans, pools, side streams) left to the collector:
    every replay reproduces the eager schedule's output and input gradients bit for bit"""
    import gc
    import itertools
    from e2_tts_pytorch_amd import Transformer
    dev, T, B = 'cuda', 24, 2
    for k, (fa, laser, fourier) in enumerate(itertools.product((False, True), repeat=3)):
        random.seed(k)
        torch.manual_seed(k)
        mod = Transformer(dim=256, depth=2, heads=2, dropout=0., max_seq_len=64, has_freq_axis=fa, freq_heads=2, attn_laser=laser,
                          attn_fourier_embed_input=fourier)
        randomize(mod, seed=k)
        mod = mod.to(dev)
        F = 3 if fa else 1

        def step(seed):
            mod.zero_grad(set_to_none=True)
            g = torch.Generator().manual_seed(seed)
            shape = (B, F, T, 256) if fa else (B, T, 256)
            x = torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
            t = torch.rand(B, generator=g).to(dev)
            txt = torch.randn(B, T, 128, generator=g).to(dev).requires_grad_(True)
            mask = (torch.arange(T)[None] < torch.tensor([T, T - 5])[:, None]).to(dev)
            R = torch.randn(*shape, generator=g).to(dev)
            out = mod(x, times=t, mask=mask, text_embed=txt)
            (out * R).sum().backward()
            return out.detach().clone(), x.grad.clone(), txt.grad.clone()

        mod.enable_plans(False)
        ref = step(3)
        mod.enable_plans(True)
        step(1), step(2)                                   # first sighting, recording
        for it in range(20):
            got = step(3)
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), (fa, laser, fourier, it)
            if it == 10:
                gc.collect()
        assert len([v for v in mod._plans.values() if not isinstance(v, str)]) == 1
        mod.__dict__['_me'] = mod                          # cyclic garbage, like a module held by a trainer object that went away
        del mod
    gc.collect()
    torch.cuda.synchronize()


def test_plan_recording_rejects_tensor_library_ops(dev):
    """a plan replays e2k calls only: a torch op doing device work inside the recorded region must raise, not vanish"""
    from e2_tts_pytorch_amd import ops
    from e2_tts_pytorch_amd.backbone import _RecordGuard
    a = torch.ones(16).to(dev)
    with _RecordGuard():
        b = a.view(4, 4)[1:]                    # views are fine
        c = torch.empty(8, device=dev)          # so are uninitialised allocations
        with pytest.raises(RuntimeError, match='inside a recorded launch plan'):
            a + 1
    assert b.shape == (3, 4) and c.numel() == 8


def test_rmsnorm_unit_offset_checkpoints_are_converted_on_load():
    """x-transformers has shipped RMSNorm with `g` = ones / gain = g (this package, the oracle) and with `g` = zeros /
    gain = g + 1; both write the same keys and shapes, so a checkpoint of the other convention would load strict=True and
    be silently wrong by +1 in every plain RMSNorm (final_norm, the text stream's norms, every norm of the duration
    predictor).  The load hook recognises it (gains near 0 instead of near 1) and converts; plain checkpoints and
    rmsnorm_convert_on_load = False are left alone"""
    from e2_tts_pytorch_amd import Transformer, E2TTS
    from e2_tts_pytorch_amd.backbone import rmsnorm_gain_convention
    random.seed(0)
    torch.manual_seed(0)
    src = Transformer(dim=256, depth=2, heads=2, max_seq_len=32)
    with torch.no_grad():
        for n, p in src.named_parameters():
            if n.endswith('.g'):
                p.add_(torch.randn_like(p) * 0.2)            # "trained" gains around 1
    sd = src.state_dict()
    gkeys = [k for k in sd if k.endswith('.g')]
    assert gkeys and rmsnorm_gain_convention(sd) == 'plain'
    other = {k: (v - 1 if k in gkeys else v.clone()) for k, v in sd.items()}       # the same model, written with gain = g + 1
    assert rmsnorm_gain_convention(other) == 'unit_offset'
    dst = Transformer(dim=256, depth=2, heads=2, max_seq_len=32)
    with pytest.warns(UserWarning, match='unit-offset'):
        dst.load_state_dict(other, strict=True)
    assert all(torch.allclose(dst.state_dict()[k], sd[k], atol=1e-6) for k in sd)
    assert all(torch.equal(other[k], sd[k] - 1) for k in gkeys)                    # (the caller's dict is untouched)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        dst.load_state_dict(sd, strict=True)                                       # plain convention: no conversion, no warning
        dst.rmsnorm_convert_on_load = False
        dst.load_state_dict(other, strict=True)                                    # opt-out: verbatim
    assert all(torch.equal(dst.state_dict()[k], other[k]) for k in gkeys)
    # through the enclosing model (prefix 'transformer.')
    m = E2TTS(transformer=dict(dim=256, depth=2, heads=2), use_vocos=False)
    msd = m.state_dict()
    mo = {k: (v - 1 if k.endswith('.g') else v) for k, v in msd.items()}
    m2 = E2TTS(transformer=dict(dim=256, depth=2, heads=2), use_vocos=False)
    with pytest.warns(UserWarning, match='unit-offset'):
        m2.load_state_dict(mo, strict=True)
    assert all(torch.allclose(m2.state_dict()[k], msd[k], atol=1e-6) for k in msd if k.endswith('.g'))


def test_dual_source_and_grouped_weight_gradients_in_the_backbone(dev):
    """weight-gradient launches of the backward schedule: the cross-condition's (D + Dt, D + Dt) and the skip projection's
    (D, 2D) gradients as ONE dual-source launch each (e2k_gemm_tn_dual_bf16), and a layer's remaining weight gradients
    (attention out / qkv / feed-forward 1 and 2 of both streams) as ONE grouped launch at the end of the layer
    (e2k_gemm_tn_group_bf16) -- against one GEMM per block / per weight: same gradients (the summation order over the token
    splits differs), and the recorded backward holds the fused calls"""
    from e2_tts_pytorch_amd import Transformer, backbone as bbm
    random.seed(0)
    torch.manual_seed(0)
    dim, depth, B, T = 256, 2, 4, 32            # (T + 32) * B = 256 token rows (x 4 streams = 1024): multiples of 64
    mod = Transformer(dim=dim, depth=depth, heads=dim // 64, dropout=0., max_seq_len=T)
    randomize(mod)
    mod = mod.to(dev)
    x0, t0, txt0 = torch.randn(B, T, dim), torch.rand(B), torch.randn(B, T, dim // 2)
    R = torch.randn(B, T, dim).to(dev)

    def grads():
        mod.zero_grad(set_to_none=True)
        mod._drop_plans()
        for _ in range(3):                      # eager, recording, replay
            mod.zero_grad(set_to_none=True)
            out = mod(x0.to(dev), times=t0.to(dev), text_embed=txt0.to(dev))
            (out * R).sum().backward()
        st = [v for v in mod._plans.values() if not isinstance(v, str) and v.bwd]
        return {n: p.grad.detach().cpu().clone() for n, p in mod.named_parameters() if p.grad is not None}, ops_names(st[0].bwd)

    old = bbm._WGRAD_DUAL, bbm._WGRAD_GROUP, bbm._WGRAD_MIN_ROWS
    try:
        bbm._WGRAD_DUAL, bbm._WGRAD_GROUP, bbm._WGRAD_MIN_ROWS = True, True, 0       # (the fused launches are for >= 1024 rows by default)
        g1, names1 = grads()
        bbm._WGRAD_DUAL, bbm._WGRAD_GROUP = False, False
        g0, names0 = grads()
    finally:
        bbm._WGRAD_DUAL, bbm._WGRAD_GROUP, bbm._WGRAD_MIN_ROWS = old
    assert names1.count('gemm_tn_dual_bf16') == depth + depth // 2 and 'gemm_tn_dual_bf16' not in names0
    assert names1.count('gemm_tn_group_bf16') == depth and 'gemm_tn_group_bf16' not in names0      # 8 weight gradients per layer = one full group
    # per layer 8 branch gradients (4 audio + 4 text); layer 0: 4 cross blocks; layer 1: 2 cross blocks (no audio_to_text) + 2 skip blocks
    assert names0.count('gemm_tn_bf16') - names1.count('gemm_tn_bf16') == 8 * depth + 4 + 2 + 2
    for n in g0:          # (fp32 atomics in some reductions make the summation order free: tolerance, not bits)
        # the weight gradients this test is about: 1e-5.  The hyper-connections' scalar and 4-element parameters are cancelling sums over all
        # tokens whose last step is an fp32 atomic: their order noise is 1e-7 of the TERMS, measured 1.7e-5 of the sum on MI355X
        tol = 1e-5 if g0[n].numel() > 64 else 2e-4
        assert rel2(g1[n], g0[n]) < tol or float(g0[n].norm()) < 1e-7, (n, rel2(g1[n], g0[n]))


@pytest.mark.parametrize('dim', [768, 1280])
def test_backbone_other_widths(dev, dim):
    """model widths besides the benchmark's: the row kernels (hyper-connections, norms, gates) are instantiated for every
    multiple of 128 the dispatch tables list (csrc/e2k_device.h E2K_ROW_DISPATCH, hc.hip HC_DISPATCH); dim 768 runs its text
    stream at 384 = 64 lanes x 2 elements x 3 chunks, dim 1280 at 640 = 64 x 2 x 5.  Forward, input and parameter gradients
    against the oracle; unsupported widths are refused at construction, not at the first kernel call"""
    from e2_tts_pytorch_amd import Transformer
    random.seed(0)
    torch.manual_seed(0)
    kw = dict(dim=dim, depth=2, heads=2, dropout=0., max_seq_len=64)
    ref = O.Transformer(**kw)
    randomize(ref)
    mod = Transformer(**kw)
    mod.load_state_dict(ref.state_dict(), strict=True)
    mod = mod.to(dev)
    B, T = (2, 12) if gpu_shapes(dev) else (1, 8)
    x, t, txt = torch.randn(B, T, dim), torch.rand(B), torch.randn(B, T, dim // 2)
    xr, xk = x.clone().requires_grad_(True), x.clone().to(dev).requires_grad_(True)
    out_r = ref(xr, times=t, text_embed=txt)
    out_r.sum().backward()
    out_k = mod(xk, times=t.to(dev), text_embed=txt.to(dev))
    out_k.sum().backward()
    assert rel2(out_k, out_r) < 2e-2 and rel2(xk.grad, xr.grad) < 5e-2
    gp = dict(ref.named_parameters())
    for n, p in mod.named_parameters():
        if p.numel() > 64 and gp[n].grad is not None and float(gp[n].grad.norm()) > 0:
            assert rel2(p.grad, gp[n].grad) < 0.2, n
    with pytest.raises(NotImplementedError):
        Transformer(dim=320, depth=2, heads=2)
