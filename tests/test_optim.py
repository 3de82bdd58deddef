"""Optimizer-side kernels (K19: global-norm clip + ADOPT + EMA over flat buffers) against oracle/optim_oracle.py."""
import pytest
import torch

from conftest import gpu_shapes

from oracle import optim_oracle as O


def _mk(seed, sizes, dev):
    torch.manual_seed(seed)
    flat = torch.randn(sum(sizes) + 8)
    return flat, sizes


@pytest.mark.parametrize('n,max_norm,wd', [(1000, 1.0, 0.), (4099, 0., 0.01), (64, 1e9, 0.)])
def test_adopt_kernel_matches_oracle(dev, n, max_norm, wd):
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    p0 = torch.randn(n)
    pr = p0.clone().requires_grad_(True)
    opt = O.Adopt([pr], lr=1e-2, weight_decay=wd)
    pk = p0.clone().to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    shadow = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    for step in range(5):
        g = torch.randn(n) * (3.0 if step % 2 else 0.1)
        pr.grad = g.clone()
        if max_norm > 0:
            O.clip_grad_norm_([pr.grad], max_norm)
        opt.step()
        gk = g.to(dev)
        gs = None
        if max_norm > 0:
            gs = torch.zeros(1, dtype=torch.float64, device=dev)
            ops.sumsq(gk, gs)
            assert abs(gs.item() - float((g.double() ** 2).sum())) <= 1e-10 * float((g.double() ** 2).sum()) + 1e-12
        # (the C ABI takes the EFFECTIVE decay: Adopt divides weight_decay by the initial lr when decoupled_wd, its default -- FusedAdopt
        #  does that division; the decay is applied on every step, the first one -- which only sets v -- included)
        ops.adopt_step(pk, gk, m, v, step, lr=1e-2, weight_decay=wd / 1e-2, max_grad_norm=max_norm, gsumsq=gs, shadow=shadow)
        assert torch.allclose(pk.cpu(), pr.detach(), rtol=2e-5, atol=2e-6), (step, (pk.cpu() - pr.detach()).abs().max())
        if wd > 0 and step == 0:
            assert torch.allclose(pk.cpu(), p0 * (1. - wd), rtol=1e-6, atol=1e-7)          # decayed although step 0 moves nothing else
        assert torch.allclose(v.cpu(), opt.v[0], rtol=2e-5, atol=1e-7)
        assert torch.allclose(m.cpu(), opt.m[0], rtol=2e-5, atol=1e-7)
        assert torch.equal(shadow.cpu(), pk.cpu().to(torch.bfloat16)) or step == 0


def test_ema_kernel(dev):
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(1)
    e, p = torch.randn(777), torch.randn(777)
    ek = e.clone().to(dev)
    ops.ema_update(ek, p.to(dev), 0.99)
    assert torch.allclose(ek.cpu(), e.lerp(p, 0.01), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('n,active_b', [(8 * 1024 + 4, True), (8 * 1024 + 4, False), (3000, True)])
def test_adopt_kernel_with_folded_ema(dev, n, active_b):
    """e2k_adopt_step_ema = e2k_adopt_step_groups followed by e2k_ema_update on the new parameters (trainer.py:275,279), in one pass:
    parameters, moments and shadow bit for bit what the two-launch sequence gives, the average to fp32 rounding; elements of the
    skipped group (ranges, active_b False) keep parameter and moments while their average still moves.  n covers whole 512-group
    blocks inside, outside and across the ranges, and a ragged tail."""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(5)
    ranges = torch.tensor([[512, 2048 + 512], [4096, 4096 + 2048 + 8]], dtype=torch.int32) if n > 4096 else torch.tensor([[8, 1000]], dtype=torch.int32)
    d = lambda t: t.clone().to(dev)
    p0, e0 = torch.randn(n), torch.randn(n)
    m0, v0 = torch.randn(n) * 0.1, torch.rand(n) + 1e-3
    A = dict(p=d(p0), m=d(m0), v=d(v0), e=d(e0))
    B = dict(p=d(p0), m=d(m0), v=d(v0), e=d(e0))
    for step in (1, 2):
        g = torch.randn(n)
        gs = torch.zeros(1, dtype=torch.float64, device=dev)
        ops.sumsq(d(g), gs)
        kw = dict(lr=1e-2, weight_decay=0.01, max_grad_norm=1.0, gsumsq=gs, ranges=ranges.to(dev), step_b=step - 1, active_b=active_b)
        ops.adopt_step(A['p'], d(g), A['m'], A['v'], step, ema=A['e'], ema_decay=0.9, **kw)
        ops.adopt_step(B['p'], d(g), B['m'], B['v'], step, **kw)
        ops.ema_update(B['e'], B['p'], 0.9)
        for k in 'pmv':
            assert torch.equal(A[k].cpu(), B[k].cpu()), (step, k)
        assert torch.allclose(A['e'].cpu(), B['e'].cpu(), rtol=1e-6, atol=1e-7), step
    inb = torch.zeros(n, dtype=torch.bool)
    for a, b in ranges.tolist():
        inb[a:b] = True
    if not active_b:
        assert torch.equal(A['p'].cpu()[inb], p0[inb]) and torch.equal(A['m'].cpu()[inb], m0[inb])
        assert not torch.equal(A['e'].cpu()[inb], e0[inb])                   # the average followed the (unchanged) parameters
    assert not torch.equal(A['p'].cpu()[~inb], p0[~inb])
    with pytest.raises(Exception):
        ops.adopt_step(A['p'], d(g), A['m'], A['v'], 1, lr=1e-2, ema=A['e'][1:], ema_decay=0.9)


@pytest.mark.late
def test_ema_folded_into_adopt_on_model(dev):
    """FusedAdopt.attach_ema: the backbone's average is moved inside the ADOPT pass on the steps on which ema.update() moves it; two
    identical models trained side by side, one folded and one not, must agree in parameters (bit for bit) and averages (fp32
    rounding) through copy steps, skipped steps (update_every 2), text-dropped steps and moving steps; a second step() without
    ema.update() raises"""
    from e2_tts_pytorch_amd import E2TTS, ops
    from e2_tts_pytorch_amd.optim import FusedAdopt, FusedEMA
    import random
    nets = []
    for fold in (True, False):
        random.seed(0)
        torch.manual_seed(0)
        model = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0., num_registers=32 if gpu_shapes(dev) else 8), use_vocos=False, cond_drop_prob=0.).to(dev)
        opt = FusedAdopt(model, lr=1e-3, max_grad_norm=1.0)
        ema = FusedEMA(model, update_after_step=1, update_every=2)
        if fold:
            opt.attach_ema(ema)
        nets.append((model, opt, ema))
    calls = []
    real = ops.ema_update
    g = torch.Generator().manual_seed(3)
    B, T = 2, 24
    try:
        ops.ema_update = lambda e, p, d: (calls.append(e.numel()), real(e, p, d))[1]
        for step, drop in enumerate([False, False, True, False, False, True, False]):
            mel = torch.randn(B, T, 100, generator=g)
            noise = dict(x0=torch.randn(B, T, 100, generator=g), times=torch.rand(B, generator=g), frac_lengths=torch.tensor([0.8, 0.9]),
                         span_rand=torch.tensor([0.1, 0.5]), drop_text_cond=drop)
            nd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in noise.items()}
            for model, opt, ema in nets:
                model(mel.to(dev), text=['hello', 'x'], _noise=nd).loss.backward()
            # two backward passes differ by the order of the gradients' fp32 atomics (on hardware and, with its worker threads, on the host
            # model too), and ADOPT's g / sqrt(v) turns that into different updates wherever a gradient is rounding noise: the second
            # model takes the first one's gradients (into its own flat gradient buffer, so both optimizers take the flat path)
            for a, b in zip(nets[0][0].parameters(), nets[1][0].parameters()):
                assert (a.grad is None) == (b.grad is None)
                if a.grad is not None:
                    b.grad.copy_(a.grad)
            counts = []
            for model, opt, ema in nets:
                calls.clear()
                opt.step()
                opt.zero_grad()
                ema.update()
                counts.append(sum(calls))
            (ma, _, ea), (mb, _, eb) = nets
            for (n, a), b in zip(ma.named_parameters(), mb.parameters()):
                assert torch.equal(a.detach().cpu(), b.detach().cpu()), (step, n)
            for (n, a), b in zip(ea.ema_model.named_parameters(), eb.ema_model.parameters()):
                assert torch.allclose(a.cpu(), b.cpu(), rtol=1e-6, atol=1e-7), (step, n)
            if step in (2, 4, 6):          # moving steps (even, past update_after_step, initted; step 2 also drops the text): the backbone's launch is gone when folded
                nflat = nets[0][0].transformer._flat.numel()
                assert counts[1] - counts[0] == nflat, (step, counts)
            else:
                assert counts[0] == counts[1], (step, counts)
    finally:
        ops.ema_update = real
    model, opt, ema = nets[0]
    assert ema.pending_decay() is None                                    # step 7: odd, nothing pending -> step() twice is fine
    mel = torch.randn(B, T, 100, generator=g).to(dev)
    for k in range(2):
        model(mel, text=['hello', 'x']).loss.backward()
        opt.step()
        opt.zero_grad()
        if k == 0:
            ema.update()                                                  # -> step 8: a moving step is pending now
    with pytest.raises(RuntimeError):
        model(mel, text=['hello', 'x']).loss.backward()
        opt.step()


@pytest.mark.late
@pytest.mark.parametrize('persist,wd', [(False, 0.), (True, 0.), (True, 0.01)])
def test_fused_adopt_on_model(dev, persist, wd):
    """FusedAdopt / FusedEMA on a small E2TTS: runs of adjacent parameters are merged, results match the oracle optimizer
    fed with the same gradients (the backbone's parameters are views of one flat buffer).  persist: the same with
    Transformer.enable_persistent_grads() -- the gradients stay attached across zero_grad and are overwritten in place"""
    from e2_tts_pytorch_amd import E2TTS
    from e2_tts_pytorch_amd.optim import FusedAdopt, FusedEMA, _runs
    import random
    random.seed(0)
    torch.manual_seed(0)
    model = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0., num_registers=32 if gpu_shapes(dev) else 8), use_vocos=False, cond_drop_prob=0.).to(dev)
    mel = torch.randn(2, 24, 100, device=dev)
    if persist:
        model.transformer.enable_persistent_grads()
    opt = FusedAdopt(model, lr=1e-3, max_grad_norm=1.0, weight_decay=wd)
    ema = FusedEMA(model, update_after_step=0, update_every=1)
    ref_params = [p.detach().cpu().clone().requires_grad_(True) for p in opt.params]
    ref = O.Adopt(ref_params, lr=1e-3, weight_decay=wd)       # (decoupled_wd=True: the decay per step is lr * wd / init_lr = wd, from step 0 on)
    for step in range(3 if wd else 2):
        if step == 2:
            opt.lr = ref.lr = 5e-4         # a scheduler moved the rate: the decoupling still divides by the INITIAL one
        out = model(mel, text=['hello world', 'x'])
        out.loss.backward()
        pairs = [(p, p.grad) for p in opt.params if p.grad is not None]
        if step == 0:
            nruns = len(_runs(pairs))
            assert nruns < len(pairs) / 4, (nruns, len(pairs))          # merged, not one launch per tensor
            tr = model.transformer                                        # and the backbone gradients are one flat buffer
            from e2_tts_pytorch_amd.optim import _grad_base
            base = _grad_base(tr._layout.slots, tr._flat.numel())
            assert base is not None and base.numel() == tr._flat.numel()
        for rp, p in zip(ref_params, opt.params):
            rp.grad = None if p.grad is None else p.grad.detach().cpu().clone()
        O.clip_grad_norm_([rp.grad for rp in ref_params if rp.grad is not None], 1.0)
        ref.step()
        opt.step()
        opt.zero_grad()
        ema.update()
        for rp, p in zip(ref_params, opt.params):
            assert torch.allclose(p.detach().cpu(), rp.detach(), rtol=1e-4, atol=1e-6), step
        tr = model.transformer
        flat_ids = {id(q) for q, _ in tr._layout.slots}
        if persist:                 # zero_grad left the backbone's views attached and dropped everything else
            assert all(q.grad is v for (q, _), v in zip(tr._layout.slots, tr._pg.views) if q.requires_grad)
            assert all(p.grad is None for p in opt.params if id(p) not in flat_ids)
        else:
            assert all(p.grad is None for p in opt.params)
    te = ema.ema_model.transformer                    # the copy has its own flat storage (one kernel launch per update)
    assert te._flat.data_ptr() != model.transformer._flat.data_ptr()
    assert all(q.data_ptr() == te._flat.data_ptr() + off * 4 for q, off in te._layout.slots)
    # EMA followed the online model (decay schedule from the oracle restatement)
    d = O.ema_decay(2, update_after_step=0)            # (schedule restatement sanity)
    assert 0. < d < 1.
    for e, p in zip(ema.ema_model.parameters(), model.parameters()):
        assert e.shape == p.shape and torch.isfinite(e).all()


def _text_stream_names(model):
    """parameters that only the text stream touches, by the reference's names"""
    out = set()
    for n, _ in model.named_parameters():
        parts = n.split('.')
        if n == 'transformer.text_registers' or n.startswith('embed_text.'):
            out.add(n)
        elif n.startswith('transformer.layers.') and parts[3] == '1':
            out.add(n)
        elif n.startswith('transformer.hyper_conns.') and parts[3] == '1':
            out.add(n)
    return out


@pytest.mark.late
@pytest.mark.parametrize('persist', [False, True])
def test_adopt_steps_are_per_parameter_and_text_is_skipped(dev, persist):
    """Adopt keeps `steps` per parameter and skips parameters without a gradient (trainer.py:183,275).  On a step whose
    classifier-free-guidance coin drops the text (e2_tts.py:1261) the text embedding has .grad None and the backbone's
    text-stream slots of the flat gradient buffer hold zeros: FusedAdopt must leave those parameters, their moments and
    their step counts alone (an update with g = 0 would keep moving them along the old momentum and decay v), while the
    backbone stays one launch.  Sequence: text, dropped, text, dropped, text -- compared with the oracle optimizer fed
    None for the text-stream parameters on the dropped steps."""
    from e2_tts_pytorch_amd import E2TTS
    from e2_tts_pytorch_amd.optim import FusedAdopt
    import random
    random.seed(0)
    torch.manual_seed(0)
    model = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0., num_registers=32 if gpu_shapes(dev) else 8), use_vocos=False, cond_drop_prob=0.).to(dev)
    if persist:
        model.transformer.enable_persistent_grads()
    opt = FusedAdopt(model, lr=1e-3, max_grad_norm=1.0)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    text_names = _text_stream_names(model)
    tr = model.transformer
    tids = opt._text_group(tr, tr.registers.device)[1] if tr._flat is not None else None
    ref_params = [p.detach().cpu().clone().requires_grad_(True) for p in opt.params]
    ref = O.Adopt(ref_params, lr=1e-3)
    B, T = 2, 24
    g = torch.Generator().manual_seed(3)
    launches = []
    for step, drop in enumerate([False, True, False, True, False]):
        mel = torch.randn(B, T, 100, generator=g).to(dev)
        noise = dict(x0=torch.randn(B, T, 100, generator=g).to(dev), times=torch.rand(B, generator=g).to(dev),
                     frac_lengths=torch.tensor([0.8, 0.9]).to(dev), span_rand=torch.tensor([0.1, 0.5]).to(dev), drop_text_cond=drop)
        model(mel, text=['hello', 'x'], _noise=noise).loss.backward()
        for n, rp, p in zip(names, ref_params, opt.params):
            rp.grad = None if (p.grad is None or (drop and n in text_names)) else p.grad.detach().cpu().clone()
            if drop and n in text_names and p.grad is not None:
                assert float(p.grad.abs().max()) == 0., n           # the flat buffer holds zeros there, not None
        O.clip_grad_norm_([rp.grad for rp in ref_params if rp.grad is not None], 1.0)
        ref.step()
        opt.step()
        opt.zero_grad()
        for n, rp, p in zip(names, ref_params, opt.params):
            assert torch.allclose(p.detach().cpu(), rp.detach(), rtol=1e-4, atol=1e-6), (step, n)
        assert opt.steps == ref.steps, step
    # the product's notion of "text stream" (ranges of the flat layout) is the reference's, by name
    tids = opt._text_group(tr, tr._flat.device)[1]
    by_id = {id(p): n for n, p in model.named_parameters()}
    assert {by_id[i] for i in tids} == {n for n in text_names if n.startswith('transformer.')}
    i_text, i_main = names.index('transformer.text_registers'), names.index('transformer.registers')
    assert opt.steps[i_text] == 3 and opt.steps[i_main] == 5 and opt.steps[names.index('embed_text.embed.weight')] == 3
    sd = opt.state_dict()
    assert sd['state'][i_text]['steps'] == 3 and sd['state'][i_main]['steps'] == 5


def test_layout_holes_stay_zero(dev):
    """the flat parameter buffer has layout holes (the bias slots of the bias-free AdaptiveRMSNorm.to_gamma rows of the
    hoisted time-conditioning block, alignment gaps, the padded tail of the fused qkv bias row).  The flat optimizer
    treats the whole buffer as one run, so every hole must keep a zero gradient and a zero value -- otherwise it becomes
    a trained parameter that the forward reads but state_dict() does not carry."""
    from e2_tts_pytorch_amd import E2TTS
    from e2_tts_pytorch_amd.optim import FusedAdopt
    import random
    random.seed(0)
    torch.manual_seed(0)
    model = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0., num_registers=32 if gpu_shapes(dev) else 8), use_vocos=False, cond_drop_prob=0.).to(dev)
    with torch.no_grad():           # zero-init gates would hide the time-conditioning gradients
        for n, p in model.named_parameters():
            if 'to_gamma' in n:
                p.add_(torch.randn_like(p) * 0.05)
    tr = model.transformer
    tr.enable_persistent_grads()
    opt = FusedAdopt(model, lr=1e-2, max_grad_norm=1.0)
    mel = torch.randn(2, 24, 100, device=dev)
    for _ in range(3):
        model(mel, text=['hello', 'x']).loss.backward()
        opt.step()
        opt.zero_grad()
    hole = torch.ones(tr._flat.numel(), dtype=torch.bool)
    for p, off in tr._layout.slots:
        hole[off:off + p.numel()] = False
    assert int(hole.sum()) > 0
    assert float(tr._pg.buf.cpu()[hole].abs().max()) == 0., 'a layout hole received a gradient'
    assert float(tr._flat.cpu()[hole].abs().max()) == 0., 'a layout hole was trained'


def test_training_loop_reduces_loss(dev):
    """the pieces of the reference trainer's step (trainer.py:263-279) together: forward, backward, clip + ADOPT, EMA.
    A fixed batch and fixed noise draws: the flow-matching loss must go down"""
    from e2_tts_pytorch_amd import E2TTS
    from e2_tts_pytorch_amd.optim import FusedAdopt, FusedEMA
    import random
    random.seed(0)
    torch.manual_seed(0)
    model = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0., num_registers=32 if gpu_shapes(dev) else 8), use_vocos=False, cond_drop_prob=0.).to(dev)
    B, T = 2, 16
    mel = torch.randn(B, T, 100, device=dev)
    noise = dict(x0=torch.randn(B, T, 100, device=dev), times=torch.tensor([0.3, 0.7], device=dev),
                 frac_lengths=torch.tensor([0.8, 0.9], device=dev), span_rand=torch.tensor([0.1, 0.5], device=dev), drop_text_cond=False)
    opt = FusedAdopt(model, lr=3e-3, max_grad_norm=1.0)
    ema = FusedEMA(model, update_after_step=0, update_every=1)
    losses = []
    for _ in range(4):
        out = model(mel, text=['hello', 'world'], _noise=noise)
        out.loss.backward()
        opt.step()
        opt.zero_grad()
        ema.update()
        losses.append(out.loss.item())
    assert all(l == l for l in losses), losses                 # finite
    assert losses[-1] < 0.9 * losses[1], losses                # (step 0 of ADOPT only initialises v)


@pytest.mark.late
def test_checkpoint_round_trip_and_format(dev, tmp_path):
    """the reference trainer's checkpoint (trainer.py:202-228): save after three steps, load into fresh objects, the
    next step is the same; the optimizer entry is a torch.optim-style dict with per-parameter 'steps' / 'm' / 'v'
    (adam_atan2_pytorch's keys) that the oracle optimizer can continue from"""
    from e2_tts_pytorch_amd import E2TTS
    from e2_tts_pytorch_amd.optim import FusedAdopt, FusedEMA, save_checkpoint, load_checkpoint
    import random

    def make():
        random.seed(0)
        torch.manual_seed(0)
        model = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0., num_registers=32 if gpu_shapes(dev) else 8), use_vocos=False, cond_drop_prob=0.).to(dev)
        return model, FusedAdopt(model, lr=1e-3, max_grad_norm=1.0), FusedEMA(model, update_after_step=0, update_every=1)

    def set_grads(model, seed):                # fixed gradients: the optimizer is then deterministic
        g = torch.Generator().manual_seed(seed)
        for p in model.parameters():
            p.grad = (torch.randn(p.shape, generator=g) * 0.01).to(dev)

    model, opt, ema = make()
    mel = torch.randn(2, 24, 100, device=dev)
    model(mel, text=['ab', 'c']).loss.backward()          # one real backward first, so the backbone's flat-gradient run exists
    opt.step()
    ema.update()
    for s in (1, 2):
        set_grads(model, s)
        opt.step()
        ema.update()
    path = tmp_path / 'ck.pt'
    save_checkpoint(path, model, opt, ema, step=3)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {'model_state_dict', 'optimizer_state_dict', 'ema_model_state_dict', 'scheduler_state_dict', 'step'}
    osd = ck['optimizer_state_dict']
    n = len(list(model.parameters()))
    assert osd['param_groups'][0]['params'] == list(range(n)) and len(osd['state']) == n
    assert all(set(st) == {'steps', 'm', 'v'} and st['steps'] == 3 for st in osd['state'].values())
    assert all(st['m'].shape == p.shape for st, p in zip(osd['state'].values(), model.parameters()))
    assert {'initted', 'step'} <= set(ck['ema_model_state_dict']) and all(
        k.startswith('ema_model.') for k in ck['ema_model_state_dict'] if k not in ('initted', 'step'))
    # (a) the oracle optimizer continues from the exported state exactly as the fused one does
    ref_params = [p.detach().cpu().clone().requires_grad_(True) for p in model.parameters()]
    ref = O.Adopt(ref_params, lr=1e-3)
    ref.steps = [3] * len(ref_params)
    for i, st in osd['state'].items():
        ref.m[i].copy_(st['m'].cpu())
        ref.v[i].copy_(st['v'].cpu())
    set_grads(model, 7)
    for rp, p in zip(ref_params, model.parameters()):
        rp.grad = p.grad.detach().cpu().clone()
    O.clip_grad_norm_([rp.grad for rp in ref_params], 1.0)
    ref.step()
    opt.step()
    ema.update()
    for rp, p in zip(ref_params, model.parameters()):
        assert torch.allclose(p.detach().cpu(), rp.detach(), rtol=1e-4, atol=1e-6)
    # (b) fresh objects + load_checkpoint take the same step
    model2, opt2, ema2 = make()
    assert load_checkpoint(path, model2, opt2, ema2, map_location=dev) == 3
    assert ema2.step == 3 and ema2.initted and opt2.step_count == 3
    set_grads(model2, 7)
    opt2.step()
    ema2.update()
    for a, b in zip(model.parameters(), model2.parameters()):
        assert torch.allclose(a.detach().cpu(), b.detach().cpu(), rtol=1e-6, atol=1e-8)
    for a, b in zip(ema.ema_model.parameters(), ema2.ema_model.parameters()):
        assert torch.allclose(a.detach().cpu(), b.detach().cpu(), rtol=1e-6, atol=1e-8)
    sd2 = opt2.state_dict()
    assert all(torch.allclose(sd2['state'][i]['v'].cpu(), opt.state_dict()['state'][i]['v'].cpu(), rtol=1e-6, atol=1e-12) for i in range(n))
