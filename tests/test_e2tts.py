"""End-to-end parity of the HIP path with the CPU oracle: E2TTS.forward (loss, pred_flow, gradients), E2TTS.sample,
DurationPredictor, MelSpec.  North-star tolerance for the bf16 path: 1e-2 (relative, stated per quantity below)."""
import random

import pytest
import torch

from conftest import gpu_shapes

from oracle import e2tts_oracle as O

from test_backbone import randomize


def install_lib(path, host_pointers):
    from emu.install import install
    install(path, host_pointers)


def rel2(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def relmax(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def test_melspec(dev):
    from e2_tts_pytorch_amd import MelSpec
    torch.manual_seed(0)
    wave = torch.randn(2, 256 * 21 + 17)
    ref = O.MelSpec()(wave)
    got = MelSpec()(wave.to(dev))
    assert got.shape == ref.shape == (2, 100, 22)
    assert (got.cpu() - ref).abs().max().item() < 2e-3      # log-mel, fp32 FFT vs torch.stft
    # the banded filterbank contraction (each htk triangle reads only its own bins) against the dense one (bands = NULL):
    # the skipped terms are exact zeros, so only the summation of zeros differs -- same values to the last place
    from e2_tts_pytorch_amd import ops
    m = MelSpec().to(dev)
    w = wave.to(dev).contiguous()
    fb = m.mel_stft.mel_scale.fb.float().contiguous()
    win = m.mel_stft.spectrogram.window.float().contiguous()
    ops.melspec(w, win, fb, 1024, 256)
    twc, tws = ops._twiddles[(str(w.device), 1024)]
    dense = torch.empty_like(got)
    ops.lib().e2k_melspec(ops._p(w), w.shape[1], ops._p(win), ops._p(fb), ops._p(twc), ops._p(tws), ops._p(dense), 2, 1024, 256, 100,
                          None, ops._stream(w))
    assert (dense.cpu() - got.cpu()).abs().max().item() < 1e-5
    bands = next(iter(ops._mel_bands.values()))[2].cpu()
    assert int((bands[:, 1] - bands[:, 0]).sum()) < 0.05 * 513 * 100 and int((bands[:, 1] - bands[:, 0]).min()) >= 1


@pytest.mark.gpu
def test_melspec_cpu_tensor_stages_through_the_device():
    """MelSpec is constructed and called on CPU tensors by the reference's HFDataset / E2Trainer (trainer.py:96,122,188):
    a CPU wave goes through the same HIP kernel (staged to the device, log-mel copied back) and comes back on the CPU"""
    import copy
    from e2_tts_pytorch_amd import MelSpec, _lib
    install_lib(None, host_pointers=False)
    torch.manual_seed(0)
    wave = torch.randn(3, 256 * 40 + 5)
    m = MelSpec()
    got = m(wave)
    assert got.device.type == 'cpu' and m.dummy.device.type == 'cpu'
    ref = O.MelSpec()(wave)
    assert got.shape == ref.shape and (got - ref).abs().max().item() < 2e-3
    assert torch.equal(got, m(wave.cuda()).cpu())                       # the very kernel the device path runs
    m2 = copy.deepcopy(m)                                               # EMA deep-copies the model (trainer.py:170)
    assert torch.equal(m2(wave[:, None, :]), got)                       # (b, 1, nw) form of the trainer's data path


def test_melspec_cpu_tensor_without_a_device_raises():
    """no HIP device, no host model: the CPU-tensor path must fail loudly, never compute on the CPU"""
    from e2_tts_pytorch_amd import MelSpec, _lib
    if torch.cuda.is_available():
        pytest.skip('a HIP device is visible')
    install_lib(None, host_pointers=False)
    with pytest.raises(_lib.E2KError):
        MelSpec()(torch.randn(1, 4096))


def test_melspec_ragged_batch(dev):
    """one launch over a zero-padded ragged batch == the reference's data path: MelSpec of every clip on its own
    (HFDataset.__getitem__, trainer.py:101-131) followed by collate_fn's zero padding (trainer.py:61-82)"""
    from e2_tts_pytorch_amd import MelSpec
    from e2_tts_pytorch_amd.data import collate_wave_fn, mel_batch
    torch.manual_seed(1)
    lens = [256 * 21 + 17, 7200, 256 * 9, 5000]
    items = [dict(wave=torch.randn(n), text='t' * (i + 1)) for i, n in enumerate(lens)]
    # reference data path, restated on the oracle
    om = O.MelSpec()
    specs = [om(it['wave'][None])[0] for it in items]                    # (100, frames_i)
    ml = torch.tensor([sp.shape[-1] for sp in specs])
    want = torch.stack([torch.nn.functional.pad(sp, (0, int(ml.max()) - sp.shape[-1])) for sp in specs])
    batch = mel_batch(collate_wave_fn(items), MelSpec().to(dev), device=dev)
    assert batch['mel'].shape == want.shape and torch.equal(batch['mel_lengths'].cpu(), ml)
    assert batch['text'] == [it['text'] for it in items] and batch['text_lengths'].tolist() == [1, 2, 3, 4]
    assert (batch['mel'].cpu() - want).abs().max().item() < 2e-3
    for i, n in enumerate(ml.tolist()):                                  # exact zeros after every clip's last frame
        assert float(batch['mel'][i, :, n:].abs().max()) == 0. if n < want.shape[-1] else True


@pytest.mark.parametrize('orig,new', [(22050, 24000), (48000, 24000), (16000, 24000), (44100, 24000), (24000, 24000)])
def test_resample_kernel(dev, orig, new):
    """data.Resample (e2k_resample_sinc; round 6: `collate_wave_fn` demanded clips at the target rate before) against the oracle's
    restatement of torchaudio.transforms.Resample as HFDataset.__getitem__ applies it (trainer.py:116-118): a single clip, a batch, and
    a zero-padded ragged batch in which every row must come out as if it had been converted alone, with exact zeros after its end"""
    from e2_tts_pytorch_amd.data import Resample
    torch.manual_seed(2)
    rs, ro = Resample(orig, new).to(dev), O.Resample(orig, new)
    x = torch.randn(3, 2000)
    want = ro(x)
    got = rs(x.to(dev))
    assert got.shape == want.shape and (got.cpu() - want).abs().max().item() < 2e-5, (got.shape, want.shape)
    one = rs(x[0].to(dev))
    assert one.shape == want[0].shape and torch.equal(one.cpu(), got[0].cpu())
    if orig == new:
        return
    lens = torch.tensor([2000, 1234, 57])
    xr = x.clone()
    for i, n in enumerate(lens.tolist()):
        xr[i, n:] = 0.
    got, nl = rs(xr.to(dev), lens=lens.to(dev))
    for i, n in enumerate(lens.tolist()):
        w = ro(x[i, :n])
        assert int(nl[i]) == w.shape[0], (i, int(nl[i]), w.shape)
        assert (got[i, :w.shape[0]].cpu() - w).abs().max().item() < 2e-5
        assert float(got[i, w.shape[0]:].abs().max()) == 0. if w.shape[0] < got.shape[1] else True


def test_data_path_resamples_foreign_rates(dev):
    """collate_wave_fn + mel_batch on rows that carry their own 'sampling_rate' (the reference's dataset rows do, trainer.py:107) ==
    HFDataset.__getitem__ clip by clip: Resample to the model's rate where it differs, MelSpec, then collate_fn's zero padding"""
    from e2_tts_pytorch_amd import MelSpec
    from e2_tts_pytorch_amd.data import collate_wave_fn, mel_batch
    torch.manual_seed(4)
    spec = [(24000, 9000), (22050, 8000), (16000, 7000), (22050, 3100)]
    items = [dict(wave=torch.randn(n) * 0.3, text='t' * (i + 1), sampling_rate=r) for i, (r, n) in enumerate(spec)]
    om = O.MelSpec()
    specs = []
    for it in items:
        w = it['wave'] if it['sampling_rate'] == 24000 else O.Resample(it['sampling_rate'], 24000)(it['wave'])
        specs.append(om(w[None])[0])
    ml = torch.tensor([sp.shape[-1] for sp in specs])
    want = torch.stack([torch.nn.functional.pad(sp, (0, int(ml.max()) - sp.shape[-1])) for sp in specs])
    batch = mel_batch(collate_wave_fn(items), MelSpec().to(dev), device=dev)
    assert torch.equal(batch['mel_lengths'].cpu(), ml) and batch['mel'].shape[:2] == want.shape[:2]
    got = batch['mel'].cpu()[..., :want.shape[-1]]
    assert (got - want).abs().max().item() < 3e-3
    assert float(batch['mel'].cpu()[..., want.shape[-1]:].abs().max()) == 0. if batch['mel'].shape[-1] > want.shape[-1] else True


def _pair(kw, seed=0, duration_predictor=None, **extra):
    from e2_tts_pytorch_amd import E2TTS
    random.seed(seed)
    torch.manual_seed(seed)
    ref = O.E2TTS(transformer=dict(**kw), cond_drop_prob=0., duration_predictor=duration_predictor, **extra)
    randomize(ref)
    model = E2TTS(transformer=dict(**kw), use_vocos=False, cond_drop_prob=0., duration_predictor=duration_predictor, **extra)
    model.load_state_dict(ref.state_dict(), strict=True)
    return ref, model


def test_e2tts_forward_backward(dev):
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    ref, model = _pair(kw)
    model = model.to(dev)
    B, T = 2, 72
    mel = torch.randn(B, T, 100)
    lens = torch.tensor([T, T - 11])
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.tensor([0.75, 0.9]),
                 span_rand=torch.tensor([0.2, 0.7]), drop_text_cond=False)
    text = ['Hello', 'Goodbye, world']
    out_r = ref(mel, text=text, lens=lens, _noise=noise)
    out_r.loss.backward()
    dn = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.to(dev), text=text, lens=lens.to(dev), _noise=dn)
    out.loss.backward()
    assert abs(out.loss.item() - out_r.loss.item()) / abs(out_r.loss.item()) < 1e-2          # loss: 1e-2 relative
    assert relmax(out.pred_flow, out_r.pred_flow) < 3e-2                                      # max-abs / max-abs
    assert rel2(out.pred_flow, out_r.pred_flow) < 1e-2                                        # rel-L2: 1e-2
    assert torch.equal(out.cond.cpu(), out_r.cond)
    for name in ('to_pred.weight', 'proj_in.weight', 'cond_proj_in.weight', 'embed_text.embed.weight',
                 'transformer.time_cond_mlp.1.weight'):
        gk = dict(model.named_parameters())[name].grad
        gr = dict(ref.named_parameters())[name].grad
        assert rel2(gk, gr) < 8e-2, (name, rel2(gk, gr))


@pytest.mark.gpu
def test_e2tts_cfg3_width():
    """the widths the headline benchmark runs (dim 1024 / text dim 512 / 16 heads), two layers, ragged batch, against
    the oracle: the D = 1024 / 512 kernel variants end to end on the hardware"""
    from e2_tts_pytorch_amd import _lib
    install_lib(None, host_pointers=False)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    kw = dict(dim=1024, depth=2, heads=16, dropout=0.)
    ref, model = _pair(kw)
    model = model.cuda()
    B, T = 2, 150
    mel = torch.randn(B, T, 100)
    lens = torch.tensor([T, T - 37])
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.tensor([0.75, 0.9]),
                 span_rand=torch.tensor([0.2, 0.7]), drop_text_cond=False)
    text = ['Hello', 'Goodbye, world']
    out_r = ref(mel, text=text, lens=lens, _noise=noise)
    out_r.loss.backward()
    dn = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.cuda(), text=text, lens=lens.cuda(), _noise=dn)
    out.loss.backward()
    assert abs(out.loss.item() - out_r.loss.item()) / abs(out_r.loss.item()) < 1e-2
    assert rel2(out.pred_flow, out_r.pred_flow) < 1e-2
    refp = dict(ref.named_parameters())
    seen = {}
    for name in ('to_pred.weight', 'proj_in.weight', 'transformer.layers.0.0.3.to_out.weight', 'transformer.layers.1.0.3.to_q.weight',
                 'transformer.layers.1.0.7.ff.0.proj.weight', 'transformer.layers.1.0.7.ff.2.bias', 'transformer.layers.1.1.2.to_v.weight',
                 'transformer.layers.1.1.4.ff.0.proj.weight', 'transformer.layers.1.1.5.text_to_audio.weight',
                 'transformer.layers.1.0.1.dw_conv1d.0.weight'):
        gk, gr = dict(model.named_parameters())[name].grad, refp[name].grad
        assert gk is not None and rel2(gk, gr) < 0.03, (name, rel2(gk, gr))          # (round 6: 0.15 before; measured 0.4-1.3 %, profiles/r06_parity_e2tts_cfg3_width.json)
        seen[name] = rel2(gk, gr)
    import json
    from pathlib import Path
    out_dir = Path(__file__).resolve().parent.parent / 'gpurun_out'
    if out_dir.is_dir():          # (the measured distances: what the 0.15 is to be tightened to)
        json.dump(dict(case='e2tts_cfg3_width', pred_flow_rel_l2=rel2(out.pred_flow, out_r.pred_flow), weight_grad_rel_l2=seen),
                  open(out_dir / 'r06_parity_e2tts_cfg3_width.json', 'w'), indent=1)


def feed_oracle_dropout_masks(ref, seed, B, N, p):
    """hand the oracle the keep masks the HIP path draws from its counter hash (oracle/dropout_hash.py): attention call of layer
    `ind`, stream t (0 audio, 1 text) uses stream id (2 ind + t) * 4, the GEGLU dropout of the same block that id + 1
    (backbone.py `_attn_block` / `_ff_block`); rows of the GEGLU mask are the token rows b * N + n"""
    from oracle.dropout_hash import attn_dropout_mask, geglu_dropout_mask
    tr = ref.transformer
    for ind, (speech, text) in enumerate(tr.layers):
        for t, (mods, ia, iff) in enumerate(((speech, 3, 7), (text, 2, 4))):
            if mods is None:
                continue
            attn, ff = mods[ia], mods[iff]
            sid = (ind * 2 + t) * 4
            attn.dropout_mask = attn_dropout_mask(seed, sid, B, attn.heads, N, p)
            F_ = ff.ff[2].in_features
            ff.ff[1].mask = geglu_dropout_mask(seed, sid + 1, B * N, F_, p).view(B, N, F_)


def spy_dropout_seed(monkeypatch):
    """records (seed, stream id) of every ops.attn_fwd call (the seed word is read back from the device in plan mode)"""
    from e2_tts_pytorch_amd import ops
    calls, orig = [], ops.attn_fwd

    def spy(st, kmask_pad, p_drop=0., seed=0, stream_id=0, seed_dev=None):
        calls.append((int(seed) if seed_dev is None else int(seed_dev.item()), int(stream_id)))
        return orig(st, kmask_pad, p_drop, seed, stream_id, seed_dev)
    monkeypatch.setattr(ops, 'attn_fwd', spy)
    return calls


def test_training_dropout_against_oracle(dev, monkeypatch):
    """one training step with dropout 0.1 (attention-probability and GEGLU dropout live in every block of both streams)
    against the fp32 oracle fed the very masks the kernels drew: loss, pred_flow, gradients"""
    kw = dict(dim=256, depth=2, heads=4, dropout=0.1)
    ref, model = _pair(kw)
    model = model.to(dev).train()
    ref.train()
    B, T = 2, 72 if gpu_shapes(dev) else 40
    N = T + 32
    mel = torch.randn(B, T, 100)
    lens = torch.tensor([T, T - 11])
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.tensor([0.75, 0.9]),
                 span_rand=torch.tensor([0.2, 0.7]), drop_text_cond=False)
    text = ['Hello', 'Goodbye, world']
    calls = spy_dropout_seed(monkeypatch)
    dn = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.to(dev), text=text, lens=lens.to(dev), _noise=dn)
    out.loss.backward()
    assert len(calls) == 4 and len({c[0] for c in calls}) == 1 and sorted(c[1] for c in calls) == [0, 4, 8, 12], calls
    feed_oracle_dropout_masks(ref, calls[0][0], B, N, 0.1)
    out_r = ref(mel, text=text, lens=lens, _noise=noise)
    out_r.loss.backward()
    assert abs(out.loss.item() - out_r.loss.item()) / abs(out_r.loss.item()) < 1e-2, (out.loss.item(), out_r.loss.item())
    assert rel2(out.pred_flow, out_r.pred_flow) < 1e-2, rel2(out.pred_flow, out_r.pred_flow)
    refp = dict(ref.named_parameters())
    seen = {}
    for name in ('to_pred.weight', 'proj_in.weight', 'transformer.layers.0.0.3.to_out.weight', 'transformer.layers.1.0.3.to_q.weight',
                 'transformer.layers.1.0.7.ff.0.proj.weight', 'transformer.layers.1.1.2.to_v.weight', 'transformer.layers.1.1.4.ff.0.proj.weight'):
        gk, gr = dict(model.named_parameters())[name].grad, refp[name].grad
        assert gk is not None and rel2(gk, gr) < 0.1, (name, rel2(gk, gr))


def test_training_dropout_shared_masks(dev):
    """a training step with dropout 0.1 (attention + GEGLU dropout active): handing the attention keep masks from the
    forward to the backward (the default) gives exactly the loss and gradients of re-hashing them in every kernel"""
    from e2_tts_pytorch_amd import E2TTS, ops
    random.seed(0)
    torch.manual_seed(0)
    model = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0.1), use_vocos=False, cond_drop_prob=0.).to(dev).train()
    B, T = 2, 40                                     # 72 positions with the registers: two key tiles
    mel = torch.randn(B, T, 100, device=dev)
    noise = dict(x0=torch.randn(B, T, 100, device=dev), times=torch.tensor([0.3, 0.7], device=dev),
                 frac_lengths=torch.tensor([0.8, 0.9], device=dev), span_rand=torch.tensor([0.1, 0.5], device=dev), drop_text_cond=False)
    res = []
    try:
        for share in (True, False):
            ops.attn_share_dropmask = share
            random.seed(1)
            torch.manual_seed(1)                     # same dropout seed draw for both passes
            model.zero_grad(set_to_none=True)
            out = model(mel, text=['ab', 'cd'], lens=torch.tensor([T, T - 15], device=dev), _noise=noise)
            out.loss.backward()
            res.append((out.loss.item(), model.to_pred.weight.grad.clone(),
                        dict(model.named_parameters())['transformer.layers.0.0.3.to_q.weight'].grad.clone()))
    finally:
        ops.attn_share_dropmask = True
    # (bit-identical in practice; the tolerance only allows for atomics-order noise should a future kernel add any on this path.
    #  A single differing keep decision would move these by orders of magnitude more.)
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0])
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-7) and torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('case', ['no_text', 'empty_string', 'short_lens', 'one_key_tile', 'text_longer_than_audio'])
def test_edge_inputs(dev, case):
    """ragged / degenerate inputs behave like the oracle: no text, an empty string in the batch, a 2-frame sample next to
    a 20-frame one, exactly one 64-position key tile, text longer than the audio (truncated)"""
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    ref, model = _pair(kw)
    model = model.to(dev)
    cfg = dict(no_text=(1, 3, None, None), empty_string=(2, 10, ['', 'hello'], None), short_lens=(2, 20, ['ab', 'cd'], [20, 2]),
               one_key_tile=(1, 32, ['x'], None), text_longer_than_audio=(1, 8, ['x' * 50], None))[case]
    B, T, text, lens = cfg
    mel = torch.randn(B, T, 100)
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.full((B,), 0.9),
                 span_rand=torch.full((B,), 0.3), drop_text_cond=False)
    kr = dict(text=text) if text is not None else {}
    kk = dict(kr)
    if lens is not None:
        kr['lens'], kk['lens'] = torch.tensor(lens), torch.tensor(lens).to(dev)
    out_r = ref(mel, _noise=noise, **kr)
    dn = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.to(dev), _noise=dn, **kk)
    out.loss.backward()
    assert torch.isfinite(out.pred_flow).all()
    assert abs(out.loss.item() - out_r.loss.item()) / abs(out_r.loss.item()) < 1e-2
    assert rel2(out.pred_flow, out_r.pred_flow) < 1.5e-2


def test_concat_cond(dev):
    """concat_cond=True (e2_tts.py:1196-1204,1263-1276): one Linear(2 * n_mels, dim) on cat(cond, x)"""
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    ref, model = _pair(kw, concat_cond=True)
    assert not hasattr(model, 'cond_proj_in') and model.proj_in.in_features == 200
    model = model.to(dev)
    B, T = 2, 40
    mel = torch.randn(B, T, 100)
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.tensor([0.8, 0.9]),
                 span_rand=torch.tensor([0.1, 0.5]), drop_text_cond=False)
    out_r = ref(mel, text=['ab', 'cd'], _noise=noise)
    out_r.loss.backward()
    dn = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.to(dev), text=['ab', 'cd'], _noise=dn)
    out.loss.backward()
    assert abs(out.loss.item() - out_r.loss.item()) / abs(out_r.loss.item()) < 1e-2
    assert rel2(out.pred_flow, out_r.pred_flow) < 1e-2
    assert rel2(model.proj_in.weight.grad, ref.proj_in.weight.grad) < 8e-2


def test_velocity_consistency_loss(dev):
    """velocity-consistency term (e2_tts.py:1558-1576) with an EMA teacher: total loss, both breakdown terms and the
    student's gradients against the oracle; the teacher receives no gradient"""
    import copy
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    ref, model = _pair(kw, velocity_consistency_weight=0.5)
    tref = copy.deepcopy(ref)
    with torch.no_grad():
        for p in tref.parameters():
            p.add_(torch.randn_like(p) * 0.02)               # a teacher that differs from the student
    from e2_tts_pytorch_amd import E2TTS
    teacher = E2TTS(transformer=dict(**kw), use_vocos=False, cond_drop_prob=0.)
    teacher.load_state_dict(tref.state_dict(), strict=True)
    model, teacher = model.to(dev), teacher.to(dev)
    B, T = 2, 48
    mel = torch.randn(B, T, 100)
    lens = torch.tensor([T, T - 9])
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.tensor([0.8, 0.9]),
                 span_rand=torch.tensor([0.1, 0.5]), drop_text_cond=False)
    text = ['Hello', 'Goodbye']
    out_r = ref(mel, text=text, lens=lens, velocity_consistency_model=tref, _noise=noise)
    out_r.loss.backward()
    dn = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.to(dev), text=text, lens=lens.to(dev), velocity_consistency_model=teacher, _noise=dn)
    out.loss.backward()
    br, bk = out_r.loss_breakdown, out.loss_breakdown
    assert br.velocity_consistency.item() > 0
    assert abs(bk.flow.item() - br.flow.item()) / br.flow.item() < 1e-2
    assert abs(bk.velocity_consistency.item() - br.velocity_consistency.item()) / br.velocity_consistency.item() < 3e-2
    assert abs(out.loss.item() - out_r.loss.item()) / out_r.loss.item() < 1e-2
    gk, gr = model.to_pred.weight.grad, ref.to_pred.weight.grad
    assert rel2(gk, gr) < 8e-2
    assert all(p.grad is None for p in teacher.parameters())


def test_e2tts_text_dropped(dev):
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    ref, model = _pair(kw, seed=1)
    model = model.to(dev)
    B, T = 1, 40
    mel = torch.randn(B, T, 100)
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.tensor([0.8]),
                 span_rand=torch.tensor([0.5]), drop_text_cond=True)
    out_r = ref(mel, text=['abc'], _noise=noise)
    dn = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in noise.items()}
    out = model(mel.to(dev), text=['abc'], _noise=dn)
    out.loss.backward()
    assert abs(out.loss.item() - out_r.loss.item()) / abs(out_r.loss.item()) < 1e-2
    # text parameters got exact zeros (DDP-friendly), not None
    g = model.transformer.layers[0][1][4].ff[2].weight.grad
    assert g is not None and float(g.abs().max()) == 0.


def test_sample(dev):
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    ref, model = _pair(kw, seed=2)
    model = model.to(dev)
    B, Tp, dur = 2, 5, 24
    cond = torch.randn(B, Tp, 100)
    y0 = torch.randn(B, dur, 100)
    text = ['Hi there', 'Yo']
    s_r = ref.sample(cond, text=text, duration=dur, steps=4, cfg_strength=1., _y0=y0)
    s = model.sample(cond.to(dev), text=text, duration=dur, steps=4, cfg_strength=1., _y0=y0.to(dev))
    assert s.shape == s_r.shape
    assert rel2(s, s_r) < 2e-2, rel2(s, s_r)


@pytest.mark.gpu
def test_cfg_passes_concurrent_bit_identical(monkeypatch):
    """sample() issues the null pass of every function evaluation on a second HIP stream next to the conditional pass
    (E2TTS._cfg_passes_concurrent; reference e2_tts.py:1303-1330 runs them one after the other).  Same kernels in the same order inside
    each pass, so the samples must be bit-identical to the sequential schedule -- with eager passes, while the plans are recorded and on
    their replays (8 steps = 14 evaluations), twice over to give a missing ordering point a chance to show, with a separate null model
    too.  Sized so that the two passes really overlap on the chip (B 8, 512 wide, 288 frames)."""
    from e2_tts_pytorch_amd import E2TTS, e2_tts
    assert torch.cuda.is_available()
    dev = 'cuda'
    random.seed(0)
    torch.manual_seed(0)
    kw = dict(dim=512, depth=4, heads=8, dropout=0.)
    model = E2TTS(transformer=dict(**kw), use_vocos=False).to(dev)
    null_model = E2TTS(transformer=dict(**kw), use_vocos=False).to(dev)
    B, dur = 8, 288
    cond = torch.randn(B, 7, 100, device=dev)
    y0 = torch.randn(B, dur, 100, device=dev)
    text = ['some text %d' % i * (1 + i % 3) for i in range(B)]
    outs = {}
    for name, conc, nm in (('seq', False, None), ('conc', True, None), ('conc2', True, None), ('seq_null', False, null_model), ('conc_null', True, null_model)):
        monkeypatch.setattr(e2_tts, '_CFG_CONCURRENT', conc)
        outs[name] = model.sample(cond, text=text, duration=dur, steps=8, cfg_strength=1.5, cfg_null_model=nm, _y0=y0)
        torch.cuda.synchronize()
    assert torch.isfinite(outs['seq']).all()
    assert torch.equal(outs['seq'], outs['conc']) and torch.equal(outs['seq'], outs['conc2'])
    assert torch.equal(outs['seq_null'], outs['conc_null']) and not torch.equal(outs['seq'], outs['seq_null'])
    assert e2_tts._CFG_STREAMS, 'the concurrent schedule did not run'


def test_sample_with_frequency_tokens(dev):
    """E2TTS(num_freq_tokens=2).sample against the oracle: the frequency axis through the classifier-free-guidance pair of
    forwards (text stream on and dropped) and the ODE steps"""
    kw = dict(dim=256, depth=2, heads=4, dropout=0., num_registers=32 if gpu_shapes(dev) else 8)
    ref, model = _pair(kw, seed=6, num_freq_tokens=2)
    model = model.to(dev)
    cond = torch.randn(1, 5, 100)
    y0 = torch.randn(1, 16, 100)
    s_r = ref.sample(cond, text=['freq'], duration=16, steps=3, cfg_strength=1., _y0=y0)
    s = model.sample(cond.to(dev), text=['freq'], duration=16, steps=3, cfg_strength=1., _y0=y0.to(dev))
    assert s.shape == s_r.shape and rel2(s, s_r) < 2e-2, rel2(s, s_r)


@pytest.mark.parametrize('method', ['euler', 'rk4'])
def test_sample_other_fixed_grid_solvers(dev, method):
    """odeint_kwargs method 'euler' / 'rk4' (torchdiffeq's fixed-grid solvers) against the oracle restatement"""
    kw = dict(dim=256, depth=2, heads=4, dropout=0., num_registers=32 if gpu_shapes(dev) else 8)
    ref, model = _pair(kw, seed=4, odeint_kwargs=dict(method=method))
    model = model.to(dev)
    cond = torch.randn(1, 5, 100)
    y0 = torch.randn(1, 20, 100)
    s_r = ref.sample(cond, text=['solver'], duration=20, steps=3, cfg_strength=0.5, _y0=y0)
    s = model.sample(cond.to(dev), text=['solver'], duration=20, steps=3, cfg_strength=0.5, _y0=y0.to(dev))
    assert rel2(s, s_r) < 2e-2, rel2(s, s_r)


def test_sample_adaptive_dopri5(dev):
    """odeint_kwargs(method='dopri5', atol, rtol) -- torchdiffeq's adaptive default, which the reference forwards to odeint
    (e2_tts.py:1122-1126,1421): the adaptive solution agrees with a fine fixed-grid midpoint integration of the SAME model
    (the HIP path's own vector field), tighter tolerance = closer"""
    from e2_tts_pytorch_amd import E2TTS
    kw = dict(dim=256, depth=2, heads=4, dropout=0., num_registers=32 if gpu_shapes(dev) else 8)
    random.seed(5)
    torch.manual_seed(5)
    model = E2TTS(transformer=dict(**kw), use_vocos=False, cond_drop_prob=0.)
    randomize(model)
    model = model.to(dev).eval()
    on_gpu = dev == 'cuda'                                             # (the host model runs ~1 s per function evaluation: coarser there)
    dur = 12 if on_gpu else 8
    cond = torch.randn(1, 5, 100).to(dev)
    y0 = torch.randn(1, dur, 100).to(dev)
    fine = model.sample(cond, text=['solver'], duration=dur, steps=17 if on_gpu else 7, cfg_strength=0., _y0=y0)
    errs = []
    for tol in ((3e-2, 2e-3) if on_gpu else (2e-2,)):
        model.odeint_kwargs = dict(method='dopri5', atol=tol, rtol=tol)
        s = model.sample(cond, text=['solver'], duration=dur, steps=5, cfg_strength=0., _y0=y0)
        errs.append(rel2(s, fine))
    model.odeint_kwargs = dict(method='midpoint')
    assert errs[-1] < (1e-2 if on_gpu else 5e-2) and errs[-1] <= errs[0] + 1e-3, errs


def test_duration_predictor(dev):
    from e2_tts_pytorch_amd import DurationPredictor
    random.seed(3)
    torch.manual_seed(3)
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    ref = O.DurationPredictor(transformer=dict(**kw))
    randomize(ref)
    model = DurationPredictor(transformer=dict(**kw))
    model.load_state_dict(ref.state_dict(), strict=True)
    model = model.to(dev)
    B, T = 2, 48
    mel = torch.randn(B, T, 100)
    lens = torch.tensor([T, 30])
    rfi = torch.tensor([0.6, 0.9])
    loss_r = ref(mel, text=['ab', 'cde'], lens=lens, _rand_frac_index=rfi)
    loss = model(mel.to(dev), text=['ab', 'cde'], lens=lens.to(dev), _rand_frac_index=rfi.to(dev))
    loss.backward()
    assert abs(loss.item() - loss_r.item()) / abs(loss_r.item()) < 2e-2
    pred_r = ref(mel, text=['ab', 'cde'], lens=lens, return_loss=False)
    with torch.no_grad():
        pred = model(mel.to(dev), text=['ab', 'cde'], lens=lens.to(dev), return_loss=False)
    assert rel2(pred, pred_r) < 2e-2


def test_flow_prologue_kernel_matches_the_tensor_library(dev, monkeypatch):
    """e2k_flow_pack (round 6): w / flow / cond of E2TTS.forward (e2_tts.py:1519-1543) and the projection's bf16 operands from one kernel.
    flow and cond bit for bit the tensor-library expressions, w's bf16 image that of `(1 - t) * x0 + t * x1`; and a training step with
    the fused prologue returns the same cond and prediction (bit for bit), loss and input-projection gradients as with E2K_FUSE_FLOW_PROLOGUE=0"""
    from e2_tts_pytorch_amd import ops
    import e2_tts_pytorch_amd.e2_tts as E
    torch.manual_seed(5)
    B, T, C = 3, 37, 100
    x0, x1, t = torch.randn(B, T, C), torch.randn(B, T, C), torch.rand(B)
    span = torch.rand(B, T) > 0.4
    wb, cb, flow, cond = ops.flow_pack(x0.to(dev), x1.to(dev), t.to(dev), span.to(dev), 104)
    w = (1. - t[:, None, None]) * x0 + t[:, None, None] * x1
    c = torch.where(span[..., None], torch.zeros_like(x1), x1)
    assert torch.equal(flow.cpu(), x1 - x0) and torch.equal(cond.cpu(), c)
    assert torch.equal(wb.cpu()[:, :C].view(B, T, C), w.to(torch.bfloat16)) and torch.equal(cb.cpu()[:, :C].view(B, T, C), c.to(torch.bfloat16))
    assert float(wb.cpu()[:, C:].float().abs().max()) == 0. and float(cb.cpu()[:, C:].float().abs().max()) == 0.
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    ref, model = _pair(kw)
    model = model.to(dev)
    mel = torch.randn(2, 40, 100).to(dev)
    noise = dict(x0=torch.randn(2, 40, 100).to(dev), times=torch.rand(2).to(dev), frac_lengths=torch.tensor([0.75, 0.9]).to(dev),
                 span_rand=torch.tensor([0.2, 0.7]).to(dev), drop_text_cond=False)
    outs = []
    for fuse in (True, False):
        monkeypatch.setattr(E, '_FUSE_FLOW_PROLOGUE', fuse)
        model.zero_grad(set_to_none=True)
        out = model(mel, text=['Hello', 'Goodbye'], _noise=noise)
        out.loss.backward()
        outs.append((out.loss.detach().cpu(), out.cond.cpu(), out.pred_flow.detach().cpu(), model.proj_in.weight.grad.cpu().clone(),
                     model.cond_proj_in.weight.grad.cpu().clone(), model.proj_in.bias.grad.cpu().clone()))
    (l0, c0, p0, *g0), (l1, c1, p1, *g1) = outs
    assert torch.equal(c0, c1) and torch.equal(p0, p1)                   # cond and the prediction: bit for bit
    assert torch.allclose(l0, l1, rtol=1e-6)                             # (the masked-MSE reduction adds its block sums with fp32 atomics: order of arrival)
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-7)


def test_duration_predictor_hl_gauss_classification(dev):
    """DurationPredictor(hl_gauss_loss=dict(...), use_regression=False) (e2_tts.py:966-967,1035-1040; round 6: refused before): the
    duration as a histogram over bins (arXiv 2403.03950) -- loss, gradient of the classification head and predictions against the
    oracle's restatement of hl_gauss_pytorch; `support` / `centers` are non-persistent, so the state dict is the regression one with
    a (num_bins, dim) head; also an independent check of the target histogram: it sums to 1 and, away from the range's ends, its mean is the target"""
    from e2_tts_pytorch_amd import DurationPredictor
    random.seed(3)
    torch.manual_seed(3)
    kw = dict(dim=256, depth=2, heads=4, dropout=0.)
    hl = dict(min_value=0., max_value=64., num_bins=24)
    ref = O.DurationPredictor(transformer=dict(**kw), hl_gauss_loss=dict(hl), use_regression=False)
    randomize(ref)
    model = DurationPredictor(transformer=dict(**kw), hl_gauss_loss=dict(hl), use_regression=False)
    assert set(model.state_dict()) == set(ref.state_dict()) and model.hl_gauss_layer.to_pred.weight.shape == (24, 256)
    model.load_state_dict(ref.state_dict(), strict=True)
    model = model.to(dev)
    B, T = 2, 48
    mel = torch.randn(B, T, 100)
    lens = torch.tensor([T, 30])
    rfi = torch.tensor([0.6, 0.9])
    loss_r = ref(mel, text=['ab', 'cde'], lens=lens, _rand_frac_index=rfi)
    loss_r.backward()
    loss = model(mel.to(dev), text=['ab', 'cde'], lens=lens.to(dev), _rand_frac_index=rfi.to(dev))
    loss.backward()
    assert abs(loss.item() - loss_r.item()) / abs(loss_r.item()) < 1e-2, (loss.item(), loss_r.item())
    assert rel2(model.hl_gauss_layer.to_pred.weight.grad, ref.hl_gauss_layer.to_pred.weight.grad) < 2e-2
    assert rel2(model.proj_in.weight.grad, ref.proj_in.weight.grad) < 5e-2
    pred_r = ref(mel, text=['ab', 'cde'], lens=lens, return_loss=False)
    with torch.no_grad():
        pred = model(mel.to(dev), text=['ab', 'cde'], lens=lens.to(dev), return_loss=False)
    assert pred.shape == (B,) and rel2(pred, pred_r) < 1e-2
    probs = model.hl_gauss_layer.hl_gauss_loss.transform_to_probs(torch.tensor([32., 47.5], device=dev))
    assert torch.allclose(probs.sum(-1).cpu(), torch.ones(2), atol=1e-5)
    assert torch.allclose((probs * model.hl_gauss_layer.hl_gauss_loss.centers).sum(-1).cpu(), torch.tensor([32., 47.5]), atol=0.05)    # (targets well inside the range: no truncated tail)
    with pytest.raises(AssertionError):
        DurationPredictor(transformer=dict(**kw), use_regression=False)          # classification without a loss definition


def test_against_golden_fixture(dev):
    """HIP path vs the committed oracle outputs (tests/golden/oracle_small.pt, made by tests/golden/make_golden.py)"""
    from pathlib import Path
    from e2_tts_pytorch_amd import E2TTS, MelSpec
    fix = torch.load(Path(__file__).resolve().parent / 'golden' / 'oracle_small.pt', weights_only=False)
    random.seed(fix['seeds'][0])
    torch.manual_seed(fix['seeds'][0])
    ref = O.E2TTS(transformer=dict(**fix['kw']), cond_drop_prob=0.)
    randomize(ref, seed=fix['seeds'][1])
    model = E2TTS(transformer=dict(**fix['kw']), use_vocos=False, cond_drop_prob=0.)
    model.load_state_dict(ref.state_dict(), strict=True)
    model = model.to(dev)
    dn = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in fix['noise'].items()}
    out = model(fix['mel'].to(dev), text=fix['text'], lens=fix['lens'].to(dev), _noise=dn)
    out.loss.backward()
    assert abs(out.loss.item() - fix['loss'].item()) / abs(fix['loss'].item()) < 1e-2
    assert rel2(out.pred_flow, fix['pred_flow']) < 1e-2
    assert rel2(model.to_pred.weight.grad, fix['grad_to_pred']) < 5e-2
    assert rel2(model.transformer.registers.grad, fix['grad_registers']) < 0.15
    assert (MelSpec()(fix['wave'].to(dev)).cpu() - fix['logmel']).abs().max().item() < 2e-3


# ---------------------------------------------------------------------------------------------- reference golden vectors
# tests/golden/reference_pinned.pt: outputs of the reference source itself (oracle/pin_against_reference.py).  The HIP
# path gets the same seeded weights and the same explicit draws; tolerances are the bf16 ones used above.

def _ref_gold():
    from pathlib import Path
    return torch.load(Path(__file__).resolve().parent / 'golden' / 'reference_pinned.pt', weights_only=False)


@pytest.mark.parametrize('case', ['e2tts_text_on', 'e2tts_cfg_drop', 'e2tts_concat_cond', 'e2tts_interp_text', 'e2tts_freq_tokens'])
def test_reference_golden_forward(dev, case):
    from e2_tts_pytorch_amd import E2TTS
    from oracle.golden_weights import fill_params
    c = _ref_gold()[case]
    random.seed(0)
    model = fill_params(E2TTS(transformer=dict(**c['kw']), use_vocos=False, cond_drop_prob=c['cond_drop_prob'], **c['extra']),
                        c['weight_seed']).to(dev)
    dn = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c['noise'].items()}
    out = model(c['mel'].to(dev), text=c['text'], lens=c['lens'].to(dev), _noise=dn)
    out.loss.backward()
    assert abs(out.loss.item() - c['loss'].item()) / abs(c['loss'].item()) < 1e-2
    assert rel2(out.pred_flow, c['pred_flow']) < 1e-2                   # north-star tolerance: 1e-2 (bf16), rel-L2
    assert relmax(out.pred_flow, c['pred_flow']) < 6e-2                 # worst single element / largest element
    assert torch.equal(out.cond.cpu(), c['cond'])
    # gradients: only their sum of magnitudes travels in the fixture, which is a fair check next to the loss (the head and
    # the input projections); element-wise gradient parity is what the oracle tests above are for
    for n in ('to_pred.weight', 'proj_in.weight', 'cond_proj_in.weight'):
        if n not in c['grad_abs_sums']:           # (concat_cond has no cond_proj_in)
            continue
        got = float(dict(model.named_parameters())[n].grad.double().abs().sum())
        # to_pred sees the backbone's forward only; the input projections hang off the backbone's INPUT gradient, which with
        # the fixture's all-random weights is ill-conditioned in the model itself (rounding weights and input to bf16 alone
        # moves it by 6 % in the fp32 oracle, tests/test_backbone.py::test_reference_golden_backbone, whose 0.15 is used here
        # too: e2tts_cfg_drop sits at 9.9 % with the two-launch GEGLU backward and 10.3 % with the fused one, the other four
        # cases within 3.5 %; a missing term moves these sums by >= 30 %)
        tol = 5e-2 if n == 'to_pred.weight' else 1.5e-1
        assert abs(got - c['grad_abs_sums'][n]) < tol * c['grad_abs_sums'][n], (n, got, c['grad_abs_sums'][n])


def test_reference_golden_sample_duration(dev):
    from e2_tts_pytorch_amd import E2TTS, DurationPredictor
    from oracle.golden_weights import fill_params
    gold = _ref_gold()
    c = gold['sample']
    random.seed(0)
    model = fill_params(E2TTS(transformer=dict(**c['kw']), use_vocos=False, cond_drop_prob=0.2), c['weight_seed']).to(dev).eval()
    out = model.sample(c['cond'].to(dev), text=c['text'], lens=c['lens'].to(dev), duration=c['duration'].to(dev), steps=c['steps'],
                       cfg_strength=c['cfg_strength'], _y0=c['y0'].to(dev))
    assert out.shape == c['out'].shape and rel2(out, c['out']) < 1e-2, rel2(out, c['out'])      # north-star tolerance (bf16)
    for key in ('duration', 'duration_freq_tokens'):              # (num_freq_tokens = 2: the frequency axis, e2_tts.py:977-1011,1098)
        c = gold[key]
        dp = fill_params(DurationPredictor(transformer=dict(**c['kw']), **c.get('extra', {})), c['weight_seed']).to(dev)
        loss = dp(c['mel'].to(dev), text=c['text'], lens=c['lens'].to(dev), _rand_frac_index=c['rand_frac_index'].to(dev))
        assert abs(loss.item() - c['loss'].item()) / abs(c['loss'].item()) < 2e-2, key


def test_reference_golden_data_path(dev):
    """ragged MelSpec kernel + collate_wave_fn / mel_batch vs what the reference's HFDataset.__getitem__ + collate_fn
    produced for the same clips (trainer.py:61-131, executed by oracle/pin_against_reference.py)"""
    from e2_tts_pytorch_amd import MelSpec
    from e2_tts_pytorch_amd.data import collate_wave_fn, mel_batch
    c = _ref_gold()['data']
    items = [dict(wave=w, text=t) for w, t in zip(c['waves'], c['text'])]
    batch = mel_batch(collate_wave_fn(items), MelSpec().to(dev), device=dev)
    assert torch.equal(batch['mel_lengths'].cpu(), c['mel_lengths']) and batch['text'] == c['text']
    assert torch.equal(batch['text_lengths'], c['text_lengths'])
    assert batch['mel'].shape == c['mel'].shape
    assert (batch['mel'].cpu() - c['mel']).abs().max().item() < 2e-3           # log-mel, fp32 FFT vs torch.stft


def test_reference_golden_sample_front_end(dev):
    """sample() front end on the HIP path vs the reference's output: raw-wave prompt (MelSpec kernel inside), duration from
    the duration predictor, max_duration clamp, autoguidance null model.  The reference drew its initial noise inside
    sample() from the CPU generator (first draw after the seed, shape of the padded prompt = shape of the output); the
    same draw is handed in through `_y0` so that the GPU run integrates from the reference's own starting point.  The
    predicted durations are truncated to integers: the output shape check holds them to the reference's"""
    from e2_tts_pytorch_amd import E2TTS
    from oracle.golden_weights import fill_params
    c = _ref_gold()['sample_front_end']
    random.seed(0)
    m = fill_params(E2TTS(transformer=dict(**c['kw']), duration_predictor=dict(transformer=dict(**c['kw'])), use_vocos=False,
                          cond_drop_prob=0.2), c['weight_seed']).to(dev).eval()
    null = fill_params(E2TTS(transformer=dict(**c['kw']), use_vocos=False, cond_drop_prob=0.2), c['null_weight_seed']).to(dev).eval()
    torch.manual_seed(c['torch_seed'])
    y0 = torch.randn(c['out'].shape)
    out = m.sample(c['wave'].to(dev), text=c['text'], lens=c['lens'].to(dev), steps=c['steps'], cfg_strength=c['cfg_strength'],
                   max_duration=c['max_duration'], cfg_null_model=null, _y0=y0.to(dev))
    assert out.shape == c['out'].shape and rel2(out, c['out']) < 1e-2, rel2(out, c['out'])      # north-star tolerance (bf16)


@pytest.mark.parametrize('remove_parallel,keep', [(True, 0.), (True, 0.3), (False, 0.)])
def test_cfg_combine_kernel(dev, remove_parallel, keep):
    """One-kernel classifier-free-guidance combine of sample() against the oracle's fp64 `project` + fp32 combine
    (e2_tts.py:113-124, 1303-1330); tolerance: a few fp32 ulps of the largest term (the kernel forms par / orth in fp64 and
    rounds each to fp32 like `project` does; only the fp64 summation order differs)."""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(3)
    pred, nul = torch.randn(3, 37, 100), torch.randn(3, 37, 100)
    nul[1] = pred[1] * 0.25                                  # a purely parallel update: orthogonal part is rounding noise only
    upd = pred - nul
    if remove_parallel:
        par, orth = O.project(upd, pred)
        upd = orth + par * keep
    ref = pred + upd * 2.0
    got = ops.cfg_combine(pred.to(dev), nul.to(dev), 2.0, keep, remove_parallel).cpu()
    assert (got - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()
    # a zero prediction: F.normalize's eps keeps the unit vector finite (0), the update passes through as orthogonal
    z = ops.cfg_combine(torch.zeros(1, 8, 100, device=dev), nul[:1, :8].contiguous().to(dev), 1.5, keep, remove_parallel).cpu()
    assert torch.equal(z, -nul[:1, :8] * 1.5)


def test_character_embed_kernel(dev):
    """CharacterEmbed as one gather kernel against the oracle module: text shorter and longer than the audio
    length, -1 padding inside the batch, gradient of the table (scatter with fp32 atomics)"""
    from e2_tts_pytorch_amd.e2_tts import CharacterEmbed
    torch.manual_seed(5)
    ref = O.CharacterEmbed(128)
    mod = CharacterEmbed(128)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(dev)
    text = O.list_str_to_tensor(['short', 'a much longer line of text than the audio has frames', ''])
    for T in (12, 70):
        R = torch.randn(3, T, 128)
        out_r = ref(text, T)
        (out_r * R).sum().backward()
        out = mod(text.to(dev), T)
        (out * R.to(dev)).sum().backward()
        assert torch.equal(out.cpu(), out_r)
        assert rel2(mod.embed.weight.grad, ref.embed.weight.grad) < 1e-6
        ref.zero_grad(), mod.zero_grad()


@pytest.mark.parametrize('with_mask', [True, False])
def test_duration_head_kernel(dev, with_mask):
    """masked mean + regression head + Softplus in one kernel against the oracle's maybe_masked_mean + HLGaussLayer:
    prediction, loss, gradients to the embedding and the head weight; one row fully masked (count clamped at 1), one
    pre-activation past Softplus' linear threshold"""
    from e2_tts_pytorch_amd.e2_tts import _DurationHeadFn
    import torch.nn.functional as F
    torch.manual_seed(6)
    B, T, D = 3, 37, 256
    embed = torch.randn(B, T, D)
    embed[2] += 3.0
    w = torch.randn(1, D) * 0.1
    w[0, :8] += 1.0                                    # row 2: z > 20
    mask = torch.rand(B, T) > 0.3 if with_mask else None
    if with_mask:
        mask[1] = False
    lens = torch.tensor([30., 12., 37.])
    er, wr = embed.clone().requires_grad_(True), w.clone().requires_grad_(True)
    pooled = O.maybe_masked_mean(er, mask)
    pred_r = F.softplus(pooled @ wr.t()).squeeze(-1)
    F.mse_loss(pred_r, lens).backward()
    ek, wk = embed.clone().to(dev).requires_grad_(True), w.clone().to(dev).requires_grad_(True)
    pred = _DurationHeadFn.apply(ek, None if mask is None else mask.to(dev), wk)
    F.mse_loss(pred, lens.to(dev)).backward()
    assert (pred.cpu() - pred_r).abs().max().item() < 1e-4 * pred_r.abs().max().item()
    assert pred_r[2].item() > 20.
    assert rel2(ek.grad, er.grad) < 1e-5 and rel2(wk.grad, wr.grad) < 1e-5
