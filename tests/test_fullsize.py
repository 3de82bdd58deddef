"""Parity at the sizes BASELINE.json names (GPU only: the CPU oracle needs seconds to a minute per case, the host model
of the kernels would need hours).  North-star tolerances for the bf16 path: loss 1e-2 relative, pred_flow rel-L2 1e-2,
sampled mel rel-L2 1e-2.  Weight gradients are reported per layer (JSON under gpurun_out/) and bounded.

  cfg1   README example exactly: E2TTS(dim=512, depth=8) (8 heads), mel = randn(2, 1024, 100), text = ['Hello', 'Goodbye'],
         one forward + backward (/root/reference/README.md:30-64) -- with the reference's own initialisation and with
         every zero-initialised path switched on (`randomize`)
  cfg3   the headline benchmark's dimensions: dim 1024, depth 24, 16 heads, T = 1024 (N = 1056 positions: 17 key tiles,
         66 row tiles and the remainder-split GEMM path the benchmark takes), B = 1
  cfg5   sample() shaped like BASELINE.json's inference config: prompt of 5 frames, 1024 target frames, 32 midpoint steps
         with classifier-free guidance (62 function evaluations x (cond + null)); small width (the ODE glue, masking,
         CFG projection and the T = 1024 attention are what this exercises)
"""
import json
import os
import random
from pathlib import Path

import pytest
import torch

from oracle import e2tts_oracle as O

from test_backbone import randomize


def install_lib(path, host_pointers):
    from emu.install import install
    install(path, host_pointers)


pytestmark = [pytest.mark.gpu, pytest.mark.late]
ROOT = Path(__file__).resolve().parent.parent


def rel2(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def _report(name, rec):
    out = ROOT / 'gpurun_out'
    if out.is_dir():
        json.dump(rec, open(out / f'r06_parity_{name}.json', 'w'), indent=1)


def _pair(kw, init, seed=0):
    from e2_tts_pytorch_amd import E2TTS, _lib
    install_lib(None, host_pointers=False)
    seed += int(os.environ.get('E2K_FULLSIZE_SEED_SHIFT', '0'))       # (diagnostic: other realisations of the same case, tools/gpu/r06w.sh)
    random.seed(seed)
    torch.manual_seed(seed)
    ref = O.E2TTS(transformer=dict(**kw), cond_drop_prob=0.)
    if init == 'randomized':
        randomize(ref)
    elif init == 'half_randomized':
        # every zero-initialised path switched on at HALF the stress strength: init + 0.5 (stress - init).  The full-strength
        # stress model is numerically exploding at depth 24 (|residual stream| reaches 9e7 in the fp32 oracle, fp16 stream
        # storage overflows to NaN, and even with fp32 streams the bf16-emulated oracle is 8 % off: the error is in every
        # bf16 GEMM operand, not in the stream format); at half strength (|stream| <= 9e3) the bf16-emulated oracle stays
        # below 1e-2 (0.86 % at dim 1024 / depth 24), and the north-star tolerance is asserted directly
        init_sd = {k: v.clone() for k, v in ref.state_dict().items()}
        randomize(ref)
        ref.load_state_dict({k: (init_sd[k] + 0.5 * (v - init_sd[k]) if v.is_floating_point() else v) for k, v in ref.state_dict().items()})
    elif init == 'trained_like':
        trained_like(ref)
    model = E2TTS(transformer=dict(**kw), use_vocos=False, cond_drop_prob=0.)
    model.load_state_dict(ref.state_dict(), strict=True)
    return ref, model.cuda()


def trained_like(model, seed=0):
    """weight statistics of a checkpoint some way into training rather than of step 0 (round 6): every
    zero-initialised projection at 0.02 randn -- the AdaLN-Zero gates with their bias moved from -2 to 0, i.e. opened to
    sigma(0) = 0.5 from 0.12 --, everything that the reference initialises non-zero left as it is.  Milder than `half_randomized`
    (which drives the residual streams to 9e3), and what the north-star tolerance is asserted on directly."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('to_gamma.weight') or 'text_to_audio' in name or 'audio_to_text' in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif name.endswith('to_gamma.bias'):
                p.zero_()
            elif 'dynamic_alpha_fn' in name or 'dynamic_beta_fn' in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif 'norm.gamma' in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)


def _train_step_parity(name, kw, B, T, text, init, flow_limit=1e-2, emulate=False):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref, model = _pair(kw, init)
    mel = torch.randn(B, T, 100)
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.full((B,), 0.85),
                 span_rand=torch.rand(B), drop_text_cond=False)
    out_r = ref(mel, text=text, _noise=noise)
    out_r.loss.backward()
    # how far the fp32 oracle itself moves when its activations are STORED in bf16 (tests/bf16_emulation.py): with every
    # zero-initialised path switched on (`randomized`) the residual streams carry large cross-condition / gate terms and
    # bf16 stream storage alone costs 2.4 % of pred_flow at depth 8 (tools/probes/flow_rounding.py) -- more than the
    # north-star tolerance.  The north-star 1e-2 is asserted for the reference's own initialisation (the configuration
    # BASELINE.json names); the stress case is held to 1.3 x this emulation.
    e_emul, g_emul = None, None
    if init == 'randomized' or emulate:
        from bf16_emulation import bf16_intermediates
        g_fp32 = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        ref.zero_grad(set_to_none=True)
        with bf16_intermediates():
            out_e = ref(mel, text=text, _noise=noise)
            out_e.loss.backward()
        e_emul = rel2(out_e.pred_flow, out_r.pred_flow)
        g_emul = {n: rel2(p.grad, g_fp32[n]) for n, p in ref.named_parameters() if p.grad is not None and float(g_fp32[n].norm()) > 0.}
        for n, p in ref.named_parameters():          # (the comparison below is against the fp32 gradients)
            p.grad = g_fp32.get(n)
    dn = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.cuda(), text=text, _noise=dn)
    out.loss.backward()
    torch.cuda.synchronize()
    e_loss = abs(out.loss.item() - out_r.loss.item()) / abs(out_r.loss.item())
    e_flow = rel2(out.pred_flow, out_r.pred_flow)
    refp = dict(ref.named_parameters())
    per_layer, worst, per_layer_emul = {}, [], {}
    for n, p in model.named_parameters():
        gr = refp[n].grad
        if gr is None or p.grad is None or float(gr.norm()) == 0.:
            continue
        e = rel2(p.grad, gr)
        key = '.'.join(n.split('.')[:3]) if n.startswith('transformer.layers.') else ('transformer.other' if n.startswith('transformer.') else 'head')
        if p.numel() >= 4096:                    # (tiny, heavily cancelling parameters are test_backbone's subject)
            per_layer.setdefault(key, []).append(e)
            worst.append((e, n))
            if g_emul is not None:
                per_layer_emul.setdefault(key, []).append(g_emul.get(n, 0.))
    worst.sort(reverse=True)
    rms = lambda v: (sum(x * x for x in v) / len(v)) ** 0.5
    layer_rms = {k: rms(v) for k, v in per_layer.items()}
    layer_rms_emul = {k: rms(v) for k, v in per_layer_emul.items()} if g_emul is not None else None
    rec = dict(case=name, init=init, kw=kw, B=B, T=T, loss=out.loss.item(), loss_ref=out_r.loss.item(), loss_rel=e_loss,
               pred_flow_rel_l2=e_flow, pred_flow_rel_l2_of_bf16_emulated_oracle=e_emul, weight_grad_rel_l2_by_layer=layer_rms,
               weight_grad_rel_l2_by_layer_of_bf16_emulated_oracle=layer_rms_emul, worst=[(round(e, 4), n) for e, n in worst[:10]],
               worst_of_bf16_emulated_oracle=([(round(g_emul.get(n, 0.), 4), n) for _, n in worst[:10]] if g_emul is not None else None))
    _report(f'{name}_{init}', rec)
    print(json.dumps({k: rec[k] for k in ('case', 'init', 'loss_rel', 'pred_flow_rel_l2')}), 'worst grads:', rec['worst'][:4],
          'emulation there:', (rec['worst_of_bf16_emulated_oracle'] or [])[:4])
    assert e_loss < 1e-2, e_loss
    assert e_flow < (flow_limit if (e_emul is None or init != 'randomized') else max(1e-2, 1.3 * e_emul)), (e_flow, e_emul)
    if emulate:
        # every tensor that is more than 3 % off the fp32 oracle must be explained by what bf16 storage does to the ORACLE's own gradient of
        # that tensor: at most 1.5 x the emulation's distance (+ 1 %).  At the reference's initialisation these are the zero-initialised
        # (D, 5) hyper-connection projections: sums over all tokens of products with a tiny upstream gradient (9.7-12 % in round 5)
        # Where bf16 storage alone puts the ORACLE more than 25 % off (trained-like weights, the last layer's two projections), the tensor's
        # gradient is rounding noise on both sides and the two distances are single draws of it.  Five realisations of the case
        # (E2K_FULLSIZE_SEED_SHIFT 0, 10 .. 40; tools/gpu/r06w.sh, MI355X): kernels 1.68 / 1.64, 0.43 / 0.46, 0.72 / 0.61, 0.81 / 0.66,
        # 0.28 / 0.14 against the emulation's 1.61 / 0.42, 0.97 / 0.96, 0.52 / 0.29, 0.51 / 0.60, 0.80 / 0.25 -- ratios from 0.35 to 3.9,
        # the two tensors moving together (they share their upstream gradient); the first draw of round 6 had been 0.28 / 0.18 before an
        # ulp-level change elsewhere in the forward moved it.  What can be held there is the size of the noise: at most twice the
        # gradient itself, or twice the emulation's distance.
        lim = lambda em: max(2.0 * em, 2.0) if em > 0.25 else 1.5 * em + 0.01
        bad = [(round(e, 4), round(g_emul.get(n, 0.), 4), n) for e, n in worst if e > 0.03 and e > lim(g_emul.get(n, 0.))]
        assert not bad, bad[:8]
    return rec


def _check_grads(rec, init, ref_limits):
    """weight gradients, rel-L2 against the fp32 oracle: per layer (rms over its tensors of >= 4096 elements) and the worst
    single tensor.  Reference initialisation: fixed limits.  Stress weights: the model itself is ill-conditioned under
    bf16 storage (at depth 24 the fp32 oracle with bf16-rounded activations is 14 % off in pred_flow and 40-50 % off in the
    early layers' weight gradients), so each layer is held to 1.5 x what that emulation already deviates by (+ 2 %)."""
    by_layer = rec['weight_grad_rel_l2_by_layer']
    if init == 'reference_init':
        assert max(by_layer.values()) < ref_limits[0], by_layer
        assert rec['worst'][0][0] < ref_limits[1], rec['worst'][:5]
        return
    emul = rec['weight_grad_rel_l2_by_layer_of_bf16_emulated_oracle']
    bad = {k: (v, emul[k]) for k, v in by_layer.items() if v > 1.5 * emul[k] + 0.02}
    assert not bad, bad


@pytest.mark.parametrize('init', ['reference_init', 'half_randomized', 'randomized'])
def test_cfg1_readme_exact(init):
    rec = _train_step_parity('cfg1', dict(dim=512, depth=8, dropout=0.), 2, 1024, ['Hello', 'Goodbye'], init)
    if init == 'half_randomized':
        assert rec['pred_flow_rel_l2'] < 1e-2, rec['pred_flow_rel_l2']       # off-init, asserted directly
        assert max(rec['weight_grad_rel_l2_by_layer'].values()) < 0.05, rec['weight_grad_rel_l2_by_layer']
        return
    _check_grads(rec, init, (0.02, 0.05))            # measured on MI355X: 0.9 % per layer, worst matrix 1.2 %


@pytest.mark.parametrize('init,B', [('reference_init', 1), ('reference_init', 2), ('half_randomized', 1), ('randomized', 1), ('trained_like', 1)])
def test_cfg3_dims_depth24(init, B):
    text = ['The quick brown fox jumps over the lazy dog.', 'Pack my box with five dozen liquor jugs!'][:B]
    # (round 6: the bf16-emulated oracle is also run at the reference's initialisation (B = 1) and for the trained-like weights, and the
    #  worst tensors are held against it -- _train_step_parity)
    rec = _train_step_parity('cfg3' if B == 1 else f'cfg3_B{B}', dict(dim=1024, depth=24, heads=16, dropout=0.), B, 1024, text, init,
                             emulate=(B == 1 and init in ('reference_init', 'trained_like')))
    if init == 'half_randomized':
        assert rec['pred_flow_rel_l2'] < 1e-2, rec['pred_flow_rel_l2']       # off-init at depth 24: the north-star tolerance, asserted directly (measured 0.88 %)
        return
    if init == 'trained_like':
        assert rec['pred_flow_rel_l2'] < 1e-2, rec['pred_flow_rel_l2']       # trained-like weight statistics: the north-star tolerance, asserted directly
        # per layer 0.9-1.4 % (bf16-emulated oracle 0.55-1.0 %); `transformer.other` holds the hyper-connection projections whose gradients
        # bf16 storage alone moves by 30 % in the oracle (12 % here), `head` the 100-channel projections (3.5 % on both sides)
        em = rec['weight_grad_rel_l2_by_layer_of_bf16_emulated_oracle']
        bad = {k: (v, em[k]) for k, v in rec['weight_grad_rel_l2_by_layer'].items() if v > max(0.03, 1.5 * em[k] + 0.01)}
        assert not bad, bad
        assert max(v for k, v in rec['weight_grad_rel_l2_by_layer'].items() if k.startswith('transformer.layers.')) < 0.03
        return
    # measured on MI355X: 0.7-1.3 % per layer; the worst single tensor is the zero-initialised hyper-connection mixing
    # projection of the last layer (12 %: a (D, 5) sum over all tokens of products with a tiny gradient), every weight
    # matrix of the attention / feed-forward / cross-condition path is below 1.5 %
    _check_grads(rec, init, (0.03, 0.15))


def test_cfg2_batch8():
    """cfg2 as BASELINE.json states it: dim 512 / depth 8, B = 8, T = 1024, synthetic mel + random text, one forward +
    backward against the CPU oracle (reference initialisation)"""
    import string
    rng = random.Random(5)
    text = [''.join(rng.choice(string.ascii_lowercase + ' ') for _ in range(rng.randint(20, 120))) for _ in range(8)]
    rec = _train_step_parity('cfg2', dict(dim=512, depth=8, dropout=0.), 8, 1024, text, 'reference_init')
    _check_grads(rec, 'reference_init', (0.02, 0.05))


def test_cfg5_sample_at_cfg3_dims():
    """sample() with the transformer of the headline config (dim 1024, depth 24, 16 heads): B = 2, prompt of 5 frames, 384
    target frames, 5 midpoint steps (8 function evaluations x (cond + null) = 16 forwards of the depth-24 backbone, no-grad
    plans, fused GEGLU epilogue) against the CPU oracle; reference initialisation"""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    kw = dict(dim=1024, depth=24, heads=16, dropout=0.)
    ref, model = _pair(kw, 'reference_init', seed=3)
    B, Tp, dur, steps = 2, 5, 384, 5
    cond = torch.randn(B, Tp, 100)
    y0 = torch.randn(B, dur, 100)
    text = ['Hi there', 'A somewhat longer line of text to speak']
    s_r = ref.sample(cond, text=text, duration=dur, steps=steps, cfg_strength=1., _y0=y0)
    s = model.sample(cond.cuda(), text=text, duration=dur, steps=steps, cfg_strength=1., _y0=y0.cuda())
    e = rel2(s, s_r)
    _report('cfg5_sample_cfg3_dims_reference_init', dict(case='sample() at cfg3 dims', kw=kw, B=B, prompt=Tp, duration=dur, steps=steps,
                                                         sampled_mel_rel_l2=e))
    print('sampled mel rel-L2 at cfg3 dims', e)
    assert s.shape == s_r.shape == (B, dur, 100) and e < 1e-2, e


@pytest.mark.parametrize('init', ['reference_init', 'randomized'])
def test_cfg5_shape_sample_32_steps(init):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    ref, model = _pair(kw, init, seed=2)
    B, Tp, dur, steps = 2, 5, 1024, 32
    cond = torch.randn(B, Tp, 100)
    y0 = torch.randn(B, dur, 100)
    text = ['Hi there', 'A somewhat longer line of text to speak']
    s_r = ref.sample(cond, text=text, duration=dur, steps=steps, cfg_strength=1., _y0=y0)
    s = model.sample(cond.cuda(), text=text, duration=dur, steps=steps, cfg_strength=1., _y0=y0.cuda())
    e = rel2(s, s_r)
    _report(f'cfg5_sample_{init}', dict(case='cfg5-shaped sample()', init=init, kw=kw, B=B, prompt=Tp, duration=dur, steps=steps,
                                        sampled_mel_rel_l2=e))
    print('sampled mel rel-L2', init, e)
    assert s.shape == s_r.shape == (B, dur, 100)
    # 31 midpoint steps x 2 evaluations integrate the per-evaluation error; north star 1e-2 for the reference's own
    # initialisation, 3e-2 for the stress weights (bf16 residual-stream storage, see _train_step_parity)
    assert e < (1e-2 if init == 'reference_init' else 3e-2), e


def test_cfg5_sample_batch8_1024_frames():
    """sample() at a full batch of full-length targets: B = 8, prompt of 5 frames, 1024 target frames, 8 midpoint steps with
    classifier-free guidance (14 function evaluations x (cond + null)), the cfg2 transformer (dim 512, depth 8: the CPU oracle
    of the depth-24 one would hold the GPU box for half an hour; its sample() is test_cfg5_sample_at_cfg3_dims)"""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    kw = dict(dim=512, depth=8, dropout=0.)
    ref, model = _pair(kw, 'reference_init', seed=4)
    B, Tp, dur, steps = 8, 5, 1024, 8
    cond = torch.randn(B, Tp, 100)
    y0 = torch.randn(B, dur, 100)
    rng = random.Random(3)
    text = [''.join(rng.choice('abcdefghijklmnopqrstuvwxyz ,.') for _ in range(rng.randint(10, 90))) for _ in range(B)]
    s_r = ref.sample(cond, text=text, duration=dur, steps=steps, cfg_strength=1., _y0=y0)
    s = model.sample(cond.cuda(), text=text, duration=dur, steps=steps, cfg_strength=1., _y0=y0.cuda())
    e = rel2(s, s_r)
    _report('cfg5_sample_B8_1024_frames_reference_init', dict(case='sample(), B = 8, 1024 frames, 8 steps', kw=kw, B=B, prompt=Tp, duration=dur,
                                                              steps=steps, sampled_mel_rel_l2=e))
    print('sampled mel rel-L2 (B = 8, 1024 frames, 8 steps)', e)
    assert s.shape == s_r.shape == (B, dur, 100) and e < 1e-2, e


@pytest.mark.parametrize('steps,rows', [(2, [0, 19, 31]), (5, [7, 24]), (9, [13])])
def test_cfg5_exact_shape_rows_against_oracle(steps, rows):
    """(round 6: also FOUR and EIGHT midpoint intervals -- steps = 5 / 9: 16 / 32 backbone forwards at B = 32, the integration error of the
    depth-24 stack accumulating over the intervals at the exact cfg5 shape, oracle on two rows / one row.)
    cfg5 as BASELINE.json states it -- B = 32, prompt of 5 frames, 1024 target frames, the cfg3 transformer (dim 1024, depth 24, 16
    heads), classifier-free guidance -- on the HIP path (no-grad launch plans at B = 32, both passes of an evaluation on two streams), one
    midpoint step (2 function evaluations x (cond + null) = 4 backbone forwards at B = 32; the CPU oracle of all 32 steps would hold the
    box for hours).  The samples of a batch do not interact (per-sample masks, per-sample CFG projection, e2_tts.py:113-124,1303-1330),
    so the fp32 oracle runs rows 0, 19 and 31 as a batch of three and each must match its row of the B = 32 result."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    kw = dict(dim=1024, depth=24, heads=16, dropout=0.)
    ref, model = _pair(kw, 'reference_init', seed=5)
    B, Tp, dur = 32, 5, 1024
    cond = torch.randn(B, Tp, 100)
    y0 = torch.randn(B, dur, 100)
    rng = random.Random(5)
    text = [''.join(rng.choice('abcdefghijklmnopqrstuvwxyz ,.') for _ in range(rng.randint(10, 90))) for _ in range(B)]
    s = model.sample(cond.cuda(), text=text, duration=dur, steps=steps, cfg_strength=1., _y0=y0.cuda())
    s_r = ref.sample(cond[rows], text=[text[r] for r in rows], duration=dur, steps=steps, cfg_strength=1., _y0=y0[rows])
    errs = [rel2(s[r], s_r[i]) for i, r in enumerate(rows)]
    _report(f'cfg5_exact_shape_rows_{steps - 1}_intervals', dict(case=f'sample() at cfg5 exactly (B 32, 1024 frames, cfg3 dims), {steps - 1} midpoint interval(s), oracle on rows', kw=kw,
                                          B=B, prompt=Tp, duration=dur, steps=steps, rows=rows, sampled_mel_rel_l2_per_row=errs))
    print('sampled mel rel-L2 per row at cfg5', errs)
    assert s.shape == (B, dur, 100) and torch.isfinite(s).all()
    assert max(errs) < 1e-2, errs
    assert rel2(s[(rows[0] + 1) % B], s_r[0]) > 0.1           # (the rows are different samples: the comparison is not vacuous)


def _grad_report(model, ref):
    refp = dict(ref.named_parameters())
    per_layer, worst = {}, []
    for n, p in model.named_parameters():
        gr = refp[n].grad
        if gr is None or p.grad is None or float(gr.norm()) == 0. or p.numel() < 4096:
            continue
        e = rel2(p.grad, gr)
        key = '.'.join(n.split('.')[:3]) if n.startswith('transformer.layers.') else ('transformer.other' if n.startswith('transformer.') else 'head')
        per_layer.setdefault(key, []).append(e)
        worst.append((e, n))
    worst.sort(reverse=True)
    rms = lambda v: (sum(x * x for x in v) / len(v)) ** 0.5
    return {k: rms(v) for k, v in per_layer.items()}, [(round(e, 4), n) for e, n in worst[:10]]


def test_cfg3_dims_dropout_against_fed_masks(monkeypatch):
    """the configuration bench.py TIMES has dropout 0.1: one training step at the headline dims (dim 1024 / depth 24 / 16 heads,
    T = 1024, B = 1) in train mode against the fp32 oracle fed the attention-probability and GEGLU keep masks the kernels drew
    (oracle/dropout_hash.py; 48 attention masks of 16 x 1056 x 1056 and 48 GEGLU masks): loss, pred_flow, per-layer weight gradients"""
    from test_e2tts import feed_oracle_dropout_masks, spy_dropout_seed
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    kw = dict(dim=1024, depth=24, heads=16, dropout=0.1)
    ref, model = _pair(kw, 'reference_init', seed=7)
    model.train()
    ref.train()
    B, T = 1, 1024
    text = ['The quick brown fox jumps over the lazy dog.']
    mel = torch.randn(B, T, 100)
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.full((B,), 0.85),
                 span_rand=torch.rand(B), drop_text_cond=False)
    calls = spy_dropout_seed(monkeypatch)
    dn = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.cuda(), text=text, _noise=dn)
    out.loss.backward()
    torch.cuda.synchronize()
    assert len(calls) == 48 and len({c[0] for c in calls}) == 1, calls[:4]
    feed_oracle_dropout_masks(ref, calls[0][0], B, T + 32, 0.1)
    out_r = ref(mel, text=text, _noise=noise)
    out_r.loss.backward()
    e_loss = abs(out.loss.item() - out_r.loss.item()) / abs(out_r.loss.item())
    e_flow = rel2(out.pred_flow, out_r.pred_flow)
    by_layer, worst = _grad_report(model, ref)
    _report('cfg3_dropout_fed_masks', dict(case='cfg3 dims, B = 1, dropout 0.1, oracle fed the kernels\' masks', kw=kw, B=B, T=T, loss_rel=e_loss,
                                           pred_flow_rel_l2=e_flow, weight_grad_rel_l2_by_layer=by_layer, worst=worst))
    print('cfg3 dropout 0.1: loss rel', e_loss, 'pred_flow rel-L2', e_flow, 'worst grads', worst[:4])
    assert e_loss < 1e-2 and e_flow < 1e-2, (e_loss, e_flow)
    assert max(by_layer.values()) < 0.03 and worst[0][0] < 0.15, (by_layer, worst[:5])


def test_cfg3_batch8_forward_backward():
    """the headline configuration at its FULL batch, forward AND backward: dim 1024 / depth 24 / 16 heads, B = 8, T = 1024, text on.
    The oracle's loss is a masked sum over samples divided by ONE global count (e2_tts.py:1578-1582), and no operation of the model
    mixes samples, so the fp32 oracle's gradients are accumulated one sample at a time: sample b alone gives loss_b = S_b / C_b,
    the batch loss is sum_b (C_b / C) loss_b (host RAM holds one sample's activations at a time).  Reference initialisation."""
    import string
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    kw = dict(dim=1024, depth=24, heads=16, dropout=0.)
    ref, model = _pair(kw, 'reference_init', seed=5)
    B, T = 8, 1024
    rng = random.Random(9)
    text = [''.join(rng.choice(string.ascii_lowercase + ' ') for _ in range(rng.randint(20, 200))) for _ in range(B)]
    mel = torch.randn(B, T, 100)
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=0.7 + 0.3 * torch.rand(B),
                 span_rand=torch.rand(B), drop_text_cond=False)
    dn = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.cuda(), text=text, _noise=dn)
    out.loss.backward()
    torch.cuda.synchronize()
    counts = (out.cond == 0).all(dim=-1).sum(dim=-1).cpu().double()          # masked frames per sample (mel is never exactly 0)
    assert float(counts.min()) > 0
    wts = (counts / counts.sum()).tolist()
    loss_r, flows = 0., []
    for b in range(B):
        nb = {k: (v[b:b + 1] if torch.is_tensor(v) else v) for k, v in noise.items()}
        out_b = ref(mel[b:b + 1], text=[text[b]], _noise=nb)
        (out_b.loss * wts[b]).backward()                                     # .grad accumulates over the samples
        loss_r += out_b.loss.item() * wts[b]
        flows.append(out_b.pred_flow.detach())
        assert torch.equal(out_b.cond, out.cond[b:b + 1].cpu())
        del out_b
    e_loss = abs(out.loss.item() - loss_r) / abs(loss_r)
    e_flow = rel2(out.pred_flow, torch.cat(flows))
    by_layer, worst = _grad_report(model, ref)
    _report('cfg3_B8_forward_backward_reference_init', dict(case='cfg3 at B = 8, forward + backward, oracle accumulated per sample', kw=kw, B=B, T=T,
                                                            loss_rel=e_loss, pred_flow_rel_l2=e_flow, weight_grad_rel_l2_by_layer=by_layer, worst=worst))
    print('cfg3 B=8 fwd+bwd: loss rel', e_loss, 'pred_flow rel-L2', e_flow, 'worst grads', worst[:4])
    assert e_loss < 1e-2 and e_flow < 1e-2, (e_loss, e_flow)
    assert max(by_layer.values()) < 0.03 and worst[0][0] < 0.15, (by_layer, worst[:5])
