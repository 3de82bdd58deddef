"""Pins the CPU oracle: (a) against the outputs of the reference's own source (tests/golden/reference_pinned.pt, written by
oracle/pin_against_reference.py -- bottom of this file); (b) for the third-party leaf modules, which the reference
imports from packages that are not installed: independent implementations available in the container, analytic
invariants of the reference initialisation, fp64 self-consistency; (c) the older oracle-only drift fixture."""
import math
import random
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

from oracle import e2tts_oracle as O
from oracle.dropout_hash import attn_dropout_mask, geglu_dropout_mask

GOLD = Path(__file__).resolve().parent / 'golden' / 'oracle_small.pt'


def test_fbank_against_hf():
    from transformers.audio_utils import mel_filter_bank
    fb = O.melscale_fbanks_htk(513, 0., 12000., 100, 24000)
    hf = torch.from_numpy(mel_filter_bank(513, 100, 0., 12000., 24000, norm=None, mel_scale='htk')).float()
    assert fb.shape == hf.shape == (513, 100)
    assert (fb - hf).abs().max().item() < 2e-5


def test_melspec_shape_and_manual_dft():
    torch.manual_seed(0)
    wave = torch.randn(1, 256 * 4)
    mel = O.MelSpec()(wave)
    assert mel.shape == (1, 100, 5)
    # frame 2 by hand: reflect pad, periodic Hann, explicit DFT in fp64
    x = F.pad(wave[:, None], (512, 512), mode='reflect')[0, 0].double()
    fr = x[2 * 256:2 * 256 + 1024] * torch.hann_window(1024, periodic=True).double()
    k = torch.arange(513).double()[:, None] * torch.arange(1024).double()[None, :]
    spec = torch.complex(torch.cos(2 * math.pi * k / 1024), -torch.sin(2 * math.pi * k / 1024)) @ torch.complex(fr, torch.zeros_like(fr))
    ref = (spec.abs().float() @ O.melscale_fbanks_htk(513, 0., 12000., 100, 24000)).clamp(min=1e-5).log()
    assert (mel[0, :, 2] - ref).abs().max().item() < 1e-3


def test_attention_against_sdpa():
    """softclamp off, no gate: the restated attention must equal torch's own SDPA"""
    torch.manual_seed(0)
    attn = O.Attention(dim=128, heads=2, dim_head=64, gate_value_heads=False, softclamp_logits=False)
    x = torch.randn(2, 17, 128)
    mask = torch.arange(17)[None] < torch.tensor([17, 11])[:, None]
    out = attn(x, mask=mask)
    q, k, v = (lin(x).view(2, 17, 2, 64).transpose(1, 2) for lin in (attn.to_q, attn.to_k, attn.to_v))
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=mask[:, None, None, :])
    ref = attn.to_out(ref.transpose(1, 2).reshape(2, 17, 128)) * mask[..., None]
    assert (out - ref).abs().max().item() < 1e-5


def test_rotary_is_a_rotation_of_adjacent_pairs():
    freqs, _ = O.RotaryEmbedding(64).forward_from_seq_len(9)
    assert freqs.shape == (1, 9, 64) and torch.equal(freqs[..., 0::2], freqs[..., 1::2])
    assert abs(freqs[0, 1, 2].item() - 10000 ** (-2 / 64)) < 1e-6
    t = torch.randn(1, 2, 9, 64)
    r = O.apply_rotary_pos_emb(t, freqs)
    assert torch.allclose(r.norm(dim=-1), t.norm(dim=-1), atol=1e-4)
    # relative-position property: <R_m q, R_n k> depends on m - n only
    q, k = torch.randn(64), torch.randn(64)
    rot = lambda v, n: O.apply_rotary_pos_emb(v.expand(1, 1, 9, 64).clone(), freqs)[0, 0, n]
    assert abs((rot(q, 5) @ rot(k, 3)).item() - (rot(q, 7) @ rot(k, 5)).item()) < 1e-3


def test_rotary_against_gptj_rotate_every_two():
    """the interleaved-pair rotary restated in the oracle (x-transformers RotaryEmbedding + apply_rotary_pos_emb, SURVEY.md
    A.6) against an independent implementation that is installed here: Hugging Face transformers' GPT-J
    (`rotate_every_two` / `apply_rotary_pos_emb` / `create_sinusoidal_positions`), same base 10000 and pair layout"""
    G = pytest.importorskip('transformers.models.gptj.modeling_gptj')
    torch.manual_seed(0)
    b, h, n, d = 2, 3, 11, 64
    t = torch.randn(b, h, n, d)
    freqs, _ = O.RotaryEmbedding(d).forward_from_seq_len(n)
    ours = O.apply_rotary_pos_emb(t, freqs)
    sincos = G.create_sinusoidal_positions(n, d)                   # (n, d): [sin | cos] halves of d / 2 frequencies
    sin, cos = sincos[:, :d // 2][None], sincos[:, d // 2:][None]
    theirs = G.apply_rotary_pos_emb(t.transpose(1, 2), sin, cos).transpose(1, 2)     # GPT-J layout (b, n, h, d)
    assert (ours - theirs).abs().max().item() < 1e-5


def test_rmsnorm_against_torch_rms_norm():
    """x-transformers RMSNorm as restated (F.normalize * sqrt(dim) * g) == torch.nn.functional.rms_norm
    with weight g (x * rsqrt(mean x^2) * g) away from the 1e-12 norm floor; AdaptiveRMSNorm with a zero-initialised
    to_gamma is the plain norm with unit gain"""
    torch.manual_seed(0)
    x = torch.randn(3, 7, 96) * 3
    m = O.RMSNorm(96)
    with torch.no_grad():
        m.g.copy_(1 + 0.3 * torch.randn(96))
    ref = F.rms_norm(x, (96,), weight=m.g, eps=0.)
    assert (m(x) - ref).abs().max().item() < 1e-5
    a = O.AdaptiveRMSNorm(96)
    assert (a(x, condition=torch.randn(3, 96)) - F.rms_norm(x, (96,), eps=0.)).abs().max().item() < 1e-5


def test_hyper_connections_against_the_papers_equations_fp64():
    """HyperConnections as restated against the dynamic hyper-connection equations of the paper
    (Zhu et al. 2024, "Hyper-Connections", eqs. for DHC with tanh), written out stream by stream in fp64 with explicit
    loops -- no einsum, no shared code with the oracle module:
        Hn_s   = norm(H_s)                                   (RMS norm with gain gamma + 1)
        B_s    = s_beta  * tanh(Hn_s . W_beta)      + B_s^static
        Am_s   = s_alpha * tanh(Hn_s . W_alpha[:,0]) + A^static[s, 0]          (width: weights of the branch input)
        Ar_s,t = s_alpha * tanh(Hn_s . W_alpha[:,1+t]) + A^static[s, 1+t]      (width: stream mixing)
        h0     = sum_s Am_s H_s                              (branch input)
        H'_t   = sum_s Ar_s,t H_s + B_t * y                  (depth: y = branch output)"""
    torch.manual_seed(0)
    S, D, b, n = 4, 32, 2, 5
    hc = O.HyperConnections(S, dim=D).double()
    with torch.no_grad():
        hc.dynamic_alpha_fn.copy_(torch.randn(D, S + 1) * 0.3)
        hc.dynamic_beta_fn.copy_(torch.randn(D) * 0.3)
        hc.dynamic_alpha_scale.fill_(0.7)
        hc.dynamic_beta_scale.fill_(0.4)
        hc.static_alpha.add_(torch.randn(S, S + 1) * 0.2)
        hc.static_beta.add_(torch.randn(S) * 0.2)
        hc.norm.gamma.copy_(torch.randn(D) * 0.2)
    H = torch.randn(b * S, n, D, dtype=torch.float64)            # the reference's layout: streams inner in the batch dim
    y = torch.randn(b, n, D, dtype=torch.float64)
    bin_o, add = hc(H)
    out_o = add(y)
    Wa, wb = hc.dynamic_alpha_fn.detach(), hc.dynamic_beta_fn.detach()
    sa, sb = float(hc.dynamic_alpha_scale.detach()), float(hc.dynamic_beta_scale.detach())
    A, B, gam = hc.static_alpha.detach(), hc.static_beta.detach(), hc.norm.gamma.detach()
    for bi in range(b):
        for ni in range(n):
            Hs = [H[bi * S + s, ni] for s in range(S)]
            Hn = [h / h.norm() * D ** 0.5 * (gam + 1) for h in Hs]
            beta = [sb * torch.tanh(Hn[s] @ wb) + B[s] for s in range(S)]
            alpha = [[sa * torch.tanh(Hn[s] @ Wa[:, t]) + A[s, t] for t in range(S + 1)] for s in range(S)]
            h0 = sum(alpha[s][0] * Hs[s] for s in range(S))
            assert (bin_o[bi, ni] - h0).abs().max().item() < 1e-12
            for t in range(S):
                Ht = sum(alpha[s][1 + t] * Hs[s] for s in range(S)) + beta[t] * y[bi, ni]
                assert (out_o[bi * S + t, ni] - Ht).abs().max().item() < 1e-12


def test_init_invariants():
    """at initialisation the zero-init paths make exact statements possible"""
    random.seed(0)
    torch.manual_seed(0)
    tr = O.Transformer(dim=128, depth=2, heads=2, dropout=0.)
    tr.eval()
    x = torch.randn(1, 10, 128)
    t = torch.rand(1)
    txt = torch.randn(1, 10, 64)
    # cross-conditioning weights are zero at init: the speech output must not depend on the text stream
    assert torch.allclose(tr(x, times=t, text_embed=txt), tr(x, times=t, text_embed=None), atol=1e-5)
    # adaptive gamma weights are zero: AdaptiveRMSNorm == plain RMSNorm with g = 1
    n = tr.layers[0][0][2]
    assert torch.allclose(n(x, condition=torch.randn(1, 128)), F.normalize(x, dim=-1) * 128 ** 0.5, atol=1e-6)
    # AdaLN-Zero gate is sigmoid(-2) for every channel at init
    g = tr.layers[0][0][5]
    assert torch.allclose(g(torch.ones(1, 3, 128), condition=torch.randn(1, 128)), torch.full((1, 3, 128), 0.11920292), atol=1e-6)
    # hyper-connections at init: branch input = one of the 4 (identical) streams, beta = 1
    hc = tr.hyper_conns[0][0][0]
    xs = O.hc_expand(x, 4)
    b, add = hc(xs)
    assert torch.allclose(b, x, atol=1e-6)
    assert torch.allclose(add(torch.zeros_like(x)), xs, atol=1e-6)
    assert torch.allclose(O.hc_reduce(add(torch.ones_like(x)), 4), 4 * (x + 1), atol=1e-5)


def test_fp64_consistency_and_grad():
    random.seed(1)
    torch.manual_seed(1)
    m = O.E2TTS(transformer=dict(dim=128, depth=2, heads=2, dropout=0.), cond_drop_prob=0.)
    mel = torch.randn(1, 12, 100)
    noise = dict(x0=torch.randn(1, 12, 100), times=torch.rand(1), frac_lengths=torch.tensor([0.9]),
                 span_rand=torch.tensor([0.1]), drop_text_cond=False)
    l32 = m(mel, text=['ab'], _noise=noise).loss
    m64 = m.double()
    n64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in noise.items()}
    l64 = m64(mel.double(), text=['ab'], _noise=n64).loss
    assert abs(l32.item() - l64.item()) < 1e-4 * abs(l64.item())
    # directional finite difference of the loss w.r.t. to_pred.weight in fp64
    l64.backward()
    w = m64.to_pred.weight
    d = torch.randn_like(w)
    eps = 1e-6
    with torch.no_grad():
        w.add_(eps * d)
        lp = m64(mel.double(), text=['ab'], _noise=n64).loss
        w.sub_(2 * eps * d)
        lm = m64(mel.double(), text=['ab'], _noise=n64).loss
        w.add_(eps * d)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd.item() - (w.grad * d).sum().item()) < 1e-5 * max(1.0, abs(fd.item()))


def test_midpoint_ode():
    """dy/dt = y on [0,1] with the explicit midpoint rule: (1 + h + h^2/2)^n"""
    t = torch.linspace(0, 1, 5)
    traj = O.odeint_midpoint(lambda _t, y: y, torch.ones(1), t)
    assert abs(traj[-1].item() - (1 + 0.25 + 0.25 ** 2 / 2) ** 4) < 1e-5


def test_fixed_grid_solvers_order():
    """euler / midpoint / rk4 (3/8 rule) on y' = -y: errors shrink with the expected orders (1, 2, 4)"""
    fn = lambda t, y: -y
    y0 = torch.ones(1, dtype=torch.float64)
    exact = math.exp(-1.0)
    errs = {}
    for m in ('euler', 'midpoint', 'rk4'):
        e = []
        for n in (9, 17):
            t = torch.linspace(0, 1, n, dtype=torch.float64)
            e.append(abs(O.odeint_fixed(fn, y0, t, m)[-1].item() - exact))
        errs[m] = e[0] / e[1]
    assert 1.8 < errs['euler'] < 2.2 and 3.6 < errs['midpoint'] < 4.4 and 14 < errs['rk4'] < 18, errs


def test_project_is_orthogonal():
    x, y = torch.randn(3, 7, 5), torch.randn(3, 7, 5)
    par, orth = O.project(x, y)
    assert torch.allclose(par + orth, x, atol=1e-5)
    assert (orth.flatten(1) * y.flatten(1)).sum(-1).abs().max().item() < 1e-4


def test_dropout_hash_statistics():
    m = attn_dropout_mask(1, 2, 1, 2, 64, 0.1)
    assert all(v == 0.0 or abs(v - 1 / 0.9) < 1e-6 for v in m.unique().tolist())
    assert abs((m == 0).float().mean().item() - 0.1) < 0.02
    g = geglu_dropout_mask(3, 4, 50, 64, 0.1)
    assert abs((g == 0).float().mean().item() - 0.1) < 0.03
    assert not torch.equal(attn_dropout_mask(1, 2, 1, 1, 32, 0.5), attn_dropout_mask(2, 2, 1, 1, 32, 0.5))
    # the attention generator draws four 16-bit samples from one hash (the second word is a cheap function of the first):
    # keep rates per position in the group, and no visible dependence between the positions of a group, along a row,
    # or between neighbouring rows
    k = (attn_dropout_mask(7, 3, 1, 1, 1024, 0.3)[0, 0] != 0).float()
    for j in range(4):
        assert abs(k[:, j::4].mean().item() - 0.7) < 0.005, j
    z = k - k.mean()
    var = (z * z).mean().item()
    for a in range(4):
        for b in range(a + 1, 4):
            assert abs((z[:, a::4] * z[:, b::4]).mean().item() / var) < 0.01, (a, b)
    assert abs((z[:, :-4] * z[:, 4:]).mean().item() / var) < 0.01 and abs((z[:-1] * z[1:]).mean().item() / var) < 0.01


def test_golden_fixture():
    fix = torch.load(GOLD, weights_only=False)
    from test_backbone import randomize
    random.seed(fix['seeds'][0])
    torch.manual_seed(fix['seeds'][0])
    model = O.E2TTS(transformer=dict(**fix['kw']), cond_drop_prob=0.)
    randomize(model, seed=fix['seeds'][1])
    wsum = sum(float(v.double().abs().sum()) for v in model.state_dict().values())
    assert abs(wsum - fix['weight_abs_sum']) < 1e-6 * fix['weight_abs_sum'], 'seeded weights differ from the fixture run'
    out = model(fix['mel'], text=fix['text'], lens=fix['lens'], _noise=fix['noise'])
    out.loss.backward()
    assert abs(out.loss.item() - fix['loss'].item()) < 1e-5
    assert (out.pred_flow - fix['pred_flow']).abs().max().item() < 1e-4
    assert (model.to_pred.weight.grad - fix['grad_to_pred']).abs().max().item() < 1e-4
    assert (O.MelSpec()(fix['wave']) - fix['logmel']).abs().max().item() < 1e-4


# ---------------------------------------------------------------------------------------------- pinned to the reference
# tests/golden/reference_pinned.pt holds outputs of /root/reference/e2_tts_pytorch/e2_tts.py ITSELF (executed by
# oracle/pin_against_reference.py with stand-ins for the uninstalled third-party leaves); weights come from seeds.

REF_GOLD = Path(__file__).resolve().parent / 'golden' / 'reference_pinned.pt'


def _maxrel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def test_reference_golden_transformer():
    from oracle.golden_weights import fill_params
    gold = torch.load(REF_GOLD, weights_only=False)
    for name in ('transformer_full', 'transformer_bare', 'transformer_variant'):
        c = gold[name]
        random.seed(0)
        m = fill_params(O.Transformer(**c['kw'], cond_on_time=c['cond_on_time']), c['weight_seed'])
        x = c['x'].clone().requires_grad_(True)
        t = c['text'].clone().requires_grad_(True) if c['text'] is not None else None
        out = m(x, times=c['times'], mask=c['mask'], text_embed=t)
        (out * c['R']).sum().backward()
        assert _maxrel(out, c['out']) < 1e-5 and _maxrel(x.grad, c['dx']) < 1e-5, name
        for n, p in m.named_parameters():
            if n in c['grad_abs_sums']:
                want = c['grad_abs_sums'][n]
                assert abs(float(p.grad.double().abs().sum()) - want) <= 1e-4 * want + 1e-9, (name, n)


def test_reference_golden_e2tts():
    from oracle.golden_weights import fill_params
    gold = torch.load(REF_GOLD, weights_only=False)
    for name in ('e2tts_text_on', 'e2tts_cfg_keep', 'e2tts_cfg_drop', 'e2tts_concat_cond', 'e2tts_interp_text'):
        c = gold[name]
        random.seed(0)
        m = fill_params(O.E2TTS(transformer=dict(**c['kw']), cond_drop_prob=c['cond_drop_prob'], **c['extra']), c['weight_seed'])
        out = m(c['mel'], text=c['text'], lens=c['lens'], _noise=c['noise'])
        out.loss.backward()
        assert abs(out.loss.item() - c['loss'].item()) <= 1e-5 * abs(c['loss'].item()), name
        assert _maxrel(out.pred_flow, c['pred_flow']) < 1e-5 and torch.equal(out.cond, c['cond']), name
        for n, p in m.named_parameters():
            if n in c['grad_abs_sums']:
                want = c['grad_abs_sums'][n]
                assert abs(float(p.grad.double().abs().sum()) - want) <= 1e-4 * want + 1e-9, (name, n)
    assert gold['e2tts_cfg_drop']['noise']['drop_text_cond'] and not gold['e2tts_cfg_keep']['noise']['drop_text_cond']
    # velocity consistency with an EMA teacher
    c = gold['velocity']
    random.seed(0)
    m = fill_params(O.E2TTS(transformer=dict(**c['kw']), cond_drop_prob=0.,
                            velocity_consistency_weight=c['velocity_consistency_weight']), c['weight_seed'])
    teacher = fill_params(O.E2TTS(transformer=dict(**c['kw']), cond_drop_prob=0.), c['teacher_weight_seed'])
    out = m(c['mel'], text=c['text'], lens=c['lens'], velocity_consistency_model=teacher, _noise=c['noise'])
    assert abs(out.loss.item() - c['loss'].item()) <= 1e-5 * abs(c['loss'].item())
    assert abs(out.loss_breakdown.velocity_consistency.item() - c['velocity_loss'].item()) <= 1e-5 * abs(c['velocity_loss'].item())
    assert c['velocity_loss'].item() > 0


def test_reference_golden_sample_and_duration():
    from oracle.golden_weights import fill_params
    gold = torch.load(REF_GOLD, weights_only=False)
    c = gold['sample']
    random.seed(0)
    m = fill_params(O.E2TTS(transformer=dict(**c['kw']), cond_drop_prob=0.2), c['weight_seed']).eval()
    out = m.sample(c['cond'], text=c['text'], lens=c['lens'], duration=c['duration'], steps=c['steps'],
                   cfg_strength=c['cfg_strength'], _y0=c['y0'])
    assert _maxrel(out, c['out']) < 1e-5
    # sample() front end: raw-wave prompt, duration from the duration predictor, max_duration clamp, autoguidance null model
    c = gold['sample_front_end']
    random.seed(0)
    m = fill_params(O.E2TTS(transformer=dict(**c['kw']), duration_predictor=dict(transformer=dict(**c['kw'])), cond_drop_prob=0.2),
                    c['weight_seed']).eval()
    null = fill_params(O.E2TTS(transformer=dict(**c['kw']), cond_drop_prob=0.2), c['null_weight_seed']).eval()
    torch.manual_seed(c['torch_seed'])
    out = m.sample(c['wave'], text=c['text'], lens=c['lens'], steps=c['steps'], cfg_strength=c['cfg_strength'],
                   max_duration=c['max_duration'], cfg_null_model=null)
    assert out.shape == c['out'].shape and _maxrel(out, c['out']) < 1e-5
    c = gold['duration']
    m = fill_params(O.DurationPredictor(transformer=dict(**c['kw'])), c['weight_seed'])
    loss = m(c['mel'], text=c['text'], lens=c['lens'], _rand_frac_index=c['rand_frac_index'])
    assert abs(loss.item() - c['loss'].item()) <= 1e-5 * abs(c['loss'].item())
    m.eval()
    with torch.no_grad():
        assert _maxrel(m(c['mel'], text=c['text'], lens=c['lens'], return_loss=False), c['pred']) < 1e-5


def test_dopri5_against_scipy_and_analytic():
    """the adaptive solver behind odeint_kwargs(method='dopri5', atol, rtol) -- torchdiffeq's default, restated from the
    published Dormand-Prince 5(4) scheme because the package is not installed -- against scipy's RK45 (the same tableau,
    an independent implementation) and analytic solutions: a decaying oscillator system and a stiff-ish scalar problem"""
    import numpy as np
    from scipy.integrate import solve_ivp
    from e2_tts_pytorch_amd.e2_tts import _odeint_dopri5
    A = torch.tensor([[-0.5, 4.0], [-4.0, -0.5]], dtype=torch.float64)
    calls = [0]

    def fn(t, y):
        calls[0] += 1
        return y @ A.T + torch.sin(3 * t) * torch.tensor([1.0, 0.0], dtype=torch.float64)
    y0 = torch.tensor([[1.0, 0.0], [0.3, -0.7]], dtype=torch.float64)
    t = torch.linspace(0, 1, 32, dtype=torch.float64)
    for tol in (1e-5, 1e-8):
        calls[0] = 0
        y = _odeint_dopri5(fn, y0, t, rtol=tol, atol=tol)
        ref = np.stack([solve_ivp(lambda tt, yy: (A.numpy() @ yy + np.sin(3 * tt) * np.array([1.0, 0.0])), (0., 1.), r.numpy(),
                                  method='RK45', rtol=1e-12, atol=1e-12).y[:, -1] for r in y0])
        assert np.abs(y.numpy() - ref).max() < 20 * tol, (tol, np.abs(y.numpy() - ref).max())
        assert calls[0] < 400                              # (adaptive: tens of steps, not thousands)
    # scalar decay with a fast rate: exact solution exp(-25 t)
    y = _odeint_dopri5(lambda t_, y_: -25. * y_, torch.ones(1, dtype=torch.float64), t, rtol=1e-7, atol=1e-9)
    assert abs(y.item() - math.exp(-25.)) < 1e-8


@pytest.mark.parametrize('orig,new', [(22050, 24000), (48000, 24000), (16000, 24000)])
def test_resample_against_an_analytic_band_limited_signal(orig, new):
    """oracle.Resample (torchaudio.transforms.Resample restated, trainer.py:116-118) is un-pinned -- torchaudio is not installed -- so it is
    held against what ANY correct resampler must do: a sum of sinusoids far below both Nyquist rates, sampled at `orig`, comes out as
    the same sinusoids sampled at `new` (away from the clip's zero-padded ends), the length is ceil(n new / orig), and an identity
    conversion returns its input"""
    import math
    n = 6000
    t0 = torch.arange(n, dtype=torch.float64) / orig
    f = [220.0, 997.0, 1810.0]
    sig = lambda t: sum(a * torch.sin(2 * math.pi * fr * t + ph) for a, fr, ph in zip((0.5, 0.3, 0.2), f, (0.1, 1.3, 2.2)))
    y = O.Resample(orig, new)(sig(t0).float())
    m = math.ceil(n * new / orig)
    assert y.shape == (m,)
    t1 = torch.arange(m, dtype=torch.float64) / new
    edge = 64
    err = (y.double() - sig(t1))[edge:-edge].abs().max().item()
    assert err < 2e-3, err
    x = torch.randn(100)
    assert torch.equal(O.Resample(24000, 24000)(x), x)
    # the filter bank: `new / gcd` phases, each a low-pass of unit DC gain (to the window's ripple)
    r = O.Resample(orig, new)
    assert r.kernel.shape[0] == new // math.gcd(orig, new) and (r.kernel.sum(-1) - 1).abs().max().item() < 2e-3
