"""E2TTS / DurationPredictor / MelSpec with the reference's constructor and call signatures
(/root/reference/e2_tts_pytorch/e2_tts.py:248-290, 956-1113, 1115-1595) on top of the HIP backbone.

The 100-channel input / output projections and the masked flow-matching loss run on the HIP GEMM / reduction kernels
(_InProjFn, _OutProjFn, _MaskedMSEFn); what stays in torch is bookkeeping around them (masks, noise draws, the
interpolation w = (1 - t) x0 + t x1, tokenisation + embedding gather): a handful of small element-wise ops.
"""
from __future__ import annotations

import os
from collections import namedtuple
from random import random

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import Module

from . import ops
from .backbone import Transformer, default, exists

# sample(): conditional and null pass of a function evaluation on two HIP streams (E2TTS._cfg_passes_concurrent); 0 = one after the other
_CFG_CONCURRENT = os.environ.get('E2K_CFG_CONCURRENT', '1') != '0'
_CFG_STREAMS = {}          # device index -> the null pass's stream

LossBreakdown = namedtuple('LossBreakdown', ['flow', 'velocity_consistency'])                       # e2_tts.py:71
E2TTSReturn = namedtuple('E2TTS', ['loss', 'cond', 'pred_flow', 'pred_data', 'loss_breakdown'])     # e2_tts.py:73


# ------------------------------------------------------------------------------------------------ host helpers

def set_if_missing_key(d, key, value):
    if key not in d:
        d.update(**{key: value})


def list_str_to_tensor(text, padding_value=-1):
    """UTF-8 byte tokenizer (e2_tts.py:128-135; like upstream the pad value is always -1)"""
    ts = [torch.tensor([*bytes(t, 'UTF-8')], dtype=torch.long) for t in text]
    return nn.utils.rnn.pad_sequence(ts, padding_value=-1, batch_first=True)


def lens_to_mask(t, length=None):                              # e2_tts.py:173-182
    if not exists(length):
        length = int(t.amax())
    seq = torch.arange(length, device=t.device)
    return seq[None, :] < t[:, None]


def mask_from_start_end_indices(seq_len, start, end):          # e2_tts.py:184-191
    max_seq_len = int(seq_len.max().item())
    seq = torch.arange(max_seq_len, device=start.device).long()
    return (seq[None, :] >= start[:, None]) & (seq[None, :] < end[:, None])


def pad_to_length(t, length, value=None):                      # e2_tts.py:226-235
    seq_len = t.shape[-1]
    if length > seq_len:
        t = F.pad(t, (0, length - seq_len), value=value)
    return t[..., :length]


def mask_from_frac_lengths(seq_len, frac_lengths, max_length=None, rand=None):       # e2_tts.py:193-210
    lengths = (frac_lengths * seq_len).long()
    max_start = seq_len - lengths
    if rand is None:
        rand = torch.rand_like(frac_lengths)
    start = (max_start * rand).long().clamp(min=0)
    end = start + lengths
    if exists(max_length):
        # same mask as mask_from_start_end_indices + pad_to_length (end <= seq_len, so positions past max(seq_len) are
        # False either way) without the reference's `.item()` device sync (e2_tts.py:189), which would stall the host
        # once per training step
        seq = torch.arange(max_length, device=start.device)
        return (seq[None, :] >= start[:, None]) & (seq[None, :] < end[:, None])
    return mask_from_start_end_indices(seq_len, start, end)


def maybe_masked_mean(t, mask=None):                           # e2_tts.py:212-224
    if not exists(mask):
        return t.mean(dim=1)
    t = torch.where(mask[..., None], t, torch.zeros_like(t))
    return t.sum(dim=1) / mask.float().sum(dim=1).clamp(min=1.)[:, None]


def project(x, y):                                             # e2_tts.py:113-124 (fp64, per flattened sample)
    shape, dtype = x.shape, x.dtype
    xf, yf = x.reshape(shape[0], -1).double(), y.reshape(shape[0], -1).double()
    unit = F.normalize(yf, dim=-1)
    parallel = (xf * unit).sum(dim=-1, keepdim=True) * unit
    orthogonal = xf - parallel
    return parallel.reshape(shape).to(dtype), orthogonal.reshape(shape).to(dtype)


# ------------------------------------------------------------------------------------------------ head / tail on the HIP GEMMs

def _r8(n):
    return (n + 7) // 8 * 8


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class _InProjFn(torch.autograd.Function):
    """a1 @ w1.T + a2 @ w2.T + bias as ONE dual-K-panel NT GEMM: `proj_in(x) + cond_proj_in(cond)`
    (e2_tts.py:1274-1277) or, for concat_cond, `proj_in(cat(cond, x))` (:1263-1272) with the weight's two column blocks.
    The 100-channel panels are padded to K = 104 in the bf16 operand copies (cast_pad kernel); weight and bias gradients
    come from the weight-gradient GEMM (bias gradient riding along as its column sums)."""

    @staticmethod
    def forward(ctx, a1, a2, w1, w2, bias):
        B, T, C = a1.shape
        M, Cp, D = B * T, _r8(C), w1.shape[0]
        a1b = ops.cast_pad_bf16(_f32c(a1.detach()).view(M, C), Cp)
        a2b = ops.cast_pad_bf16(_f32c(a2.detach()).view(M, C), Cp)
        wb = torch.empty((D, 2 * Cp), dtype=torch.bfloat16, device=a1.device)
        ops.cast_pad_bf16(w1.detach(), Cp, out=wb, col0=0)
        ops.cast_pad_bf16(w2.detach(), Cp, out=wb, col0=Cp)
        out = ops.gemm_nt(a1b, wb, a2=a2b, bias=_f32c(bias.detach()), out_dtype=torch.float32)
        ctx.save_for_backward(a1b, a2b, w1, w2)
        ctx.dims = (B, T, C, D)
        ctx.in_dtypes = (a1.dtype, a2.dtype)
        return out.view(B, T, D)

    @staticmethod
    def backward(ctx, dout):
        a1b, a2b, w1, w2 = ctx.saved_tensors
        B, T, C, D = ctx.dims
        M, dev = B * T, dout.device
        dob = ops.cast_bf16(_f32c(dout).view(-1), torch.empty((M, D), dtype=torch.bfloat16, device=dev))
        dw1, dw2, db = ops.zeros((D, C), torch.float32, dev), ops.zeros((D, C), torch.float32, dev), ops.zeros((D,), torch.float32, dev)
        ops.gemm_tn(dob, a1b[:, :C], dw1, colsum=db)
        ops.gemm_tn(dob, a2b[:, :C], dw2)
        # input gradients (nn.Linear propagates them: a trainable module in front of cond, input-gradient objectives);
        # the training step itself never asks for them
        das = [None, None]
        for i, w in enumerate((w1, w2)):
            if ctx.needs_input_grad[i]:
                wT = ops.cast_transpose_bf16(_f32c(w.detach()), torch.empty((C, D), dtype=torch.bfloat16, device=dev))
                das[i] = ops.gemm_nt(dob, wT, out_dtype=torch.float32).view(B, T, C).to(ctx.in_dtypes[i])
        return das[0], das[1], dw1, dw2, db


class _FlowInProjFn(torch.autograd.Function):
    """E2TTS.forward's prologue + input projection (e2_tts.py:1519-1543,1274-1277; round 6): ONE kernel forms
    w = (1 - t) x0 + t x1, flow = x1 - x0, cond = where(span, 0, x1) and the bf16 K-padded GEMM operands of w and cond (ten
    tensor-library launches and two cast kernels before), then the dual-K-panel GEMM of _InProjFn.  -> (h (B, T, D), flow, cond).
    For inputs that need no gradient (the training step: mel, noise and times are data); anything else takes the unfused path."""

    @staticmethod
    def forward(ctx, x0, x1, times, span_mask, w1, w2, bias):
        B, T, C = x1.shape
        Cp, D, dev = _r8(C), w1.shape[0], x1.device
        a1b, a2b, flow, cond = ops.flow_pack(_f32c(x0), _f32c(x1), _f32c(times), span_mask.contiguous(), Cp)
        wb = torch.empty((D, 2 * Cp), dtype=torch.bfloat16, device=dev)
        ops.cast_pad_bf16(w1.detach(), Cp, out=wb, col0=0)
        ops.cast_pad_bf16(w2.detach(), Cp, out=wb, col0=Cp)
        out = ops.gemm_nt(a1b, wb, a2=a2b, bias=_f32c(bias.detach()), out_dtype=torch.float32)
        ctx.save_for_backward(a1b, a2b)
        ctx.dims = (B, T, C, D)
        ctx.mark_non_differentiable(flow, cond)
        return out.view(B, T, D), flow, cond

    @staticmethod
    def backward(ctx, dout, _dflow, _dcond):
        a1b, a2b = ctx.saved_tensors
        B, T, C, D = ctx.dims
        M, dev = B * T, dout.device
        dob = ops.cast_bf16(_f32c(dout).view(-1), torch.empty((M, D), dtype=torch.bfloat16, device=dev))
        dw1, dw2, db = ops.zeros((D, C), torch.float32, dev), ops.zeros((D, C), torch.float32, dev), ops.zeros((D,), torch.float32, dev)
        ops.gemm_tn(dob, a1b[:, :C], dw1, colsum=db)
        ops.gemm_tn(dob, a2b[:, :C], dw2)
        return None, None, None, None, dw1, dw2, db


class _OutProjFn(torch.autograd.Function):
    """to_pred (e2_tts.py:1296): (B, T, D) -> (B, T, C) on the NT GEMM; backward dgrad over the K = 104 padded gradient"""

    @staticmethod
    def forward(ctx, e, w, bias):
        B, T, D = e.shape
        M, C, dev = B * T, w.shape[0], e.device
        eb = ops.cast_bf16(_f32c(e.detach()).view(-1), torch.empty((M, D), dtype=torch.bfloat16, device=dev))
        wf = _f32c(w.detach())
        wb = ops.cast_bf16(wf.view(-1), torch.empty((C, D), dtype=torch.bfloat16, device=dev))
        out = ops.gemm_nt(eb, wb, bias=_f32c(bias.detach()), out_dtype=torch.float32)
        ctx.save_for_backward(eb, wf)
        ctx.dims = (B, T, D, C)
        return out.view(B, T, C)

    @staticmethod
    def backward(ctx, dout):
        eb, wf = ctx.saved_tensors
        B, T, D, C = ctx.dims
        M, Cp, dev = B * T, _r8(C), dout.device
        dpb = ops.cast_pad_bf16(_f32c(dout).view(M, C), Cp)
        wT = ops.zeros((D, Cp), torch.bfloat16, dev)
        ops.cast_transpose_bf16(wf, wT[:, :C])
        de = ops.gemm_nt(dpb, wT, out_dtype=torch.float32)
        dw, db = ops.zeros((C, D), torch.float32, dev), ops.zeros((C,), torch.float32, dev)
        ops.gemm_tn(dpb[:, :C], eb, dw, colsum=db)
        return de.view(B, T, D), dw, db


class _MaskedMSEFn(torch.autograd.Function):
    """mean of (pred - flow)^2 over the masked span (e2_tts.py:1578-1582) in one reduction kernel, no boolean-index gather"""

    @staticmethod
    def forward(ctx, pred, flow, mask):
        B, T, C = pred.shape
        p2, f2 = _f32c(pred.detach()).view(B * T, C), _f32c(flow.detach()).view(B * T, C)
        m8 = mask.contiguous().view(torch.uint8).view(-1)
        acc = ops.masked_mse_fwd(p2, f2, m8)
        ctx.save_for_backward(p2, f2, m8, acc)
        ctx.shape = pred.shape
        return acc[0] / (acc[1] * C)

    @staticmethod
    def backward(ctx, dloss):
        p2, f2, m8, acc = ctx.saved_tensors
        dpred = ops.masked_mse_bwd(p2, f2, m8, acc, _f32c(dloss).view(1)).view(ctx.shape)
        return dpred, (-dpred if ctx.needs_input_grad[1] else None), None          # d/d(flow) = -d/d(pred)


_CHECK_TOKEN_IDS = __import__('os').environ.get('E2K_CHECK_TOKEN_IDS', '0') == '1'
_FUSE_FLOW_PROLOGUE = __import__('os').environ.get('E2K_FUSE_FLOW_PROLOGUE', '1') != '0'


def _on_kernels(t):
    return t.is_cuda or ops.host_ok()


class _CharEmbedFn(torch.autograd.Function):
    """CharacterEmbed (e2_tts.py:400-412): shift the byte tokens by one, cut / pad to the audio length with the padding row,
    gather -- one kernel; the backward scatters into the (num_embeds + 1, dim) table with fp32 atomics"""

    @staticmethod
    def forward(ctx, text, W, max_seq_len):
        # nn.Embedding raises on ids outside the table (a tokenizer that does not match text_num_embeds); the kernel only keeps
        # its reads in bounds.  Ids are checked where that is free (token tensors that are still on the host) and, for device
        # tensors, when E2K_CHECK_TOKEN_IDS=1 (one device -> host synchronisation per call)
        if text.numel() and (not text.is_cuda or _CHECK_TOKEN_IDS):
            lo, hi = int(text.min()), int(text.max())
            if lo < -1 or hi + 1 >= W.shape[0]:
                raise IndexError(f'token ids span [{lo}, {hi}] but the embedding table has {W.shape[0]} rows (ids are shifted by one; -1 = padding)')
        out, tk = ops.char_embed_fwd(text.contiguous(), W.detach().contiguous(), max_seq_len)
        ctx.save_for_backward(tk)
        ctx.V = W.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        tk, = ctx.saved_tensors
        return None, ops.char_embed_bwd(tk, _f32c(dout), ctx.V), None


class _DurationHeadFn(torch.autograd.Function):
    """maybe_masked_mean + HLGaussLayer's regression head with Softplus (e2_tts.py:1098-1111,212-224): pred (B,)"""

    @staticmethod
    def forward(ctx, embed, mask, w):
        e = _f32c(embed.detach())
        m8 = mask.contiguous().view(torch.uint8) if exists(mask) else None
        wv = w.detach().reshape(-1).contiguous()
        pred, pooled, z = ops.duration_head_fwd(e, m8, wv)
        ctx.save_for_backward(pooled, z, wv, *((m8,) if exists(m8) else ()))
        ctx.T, ctx.wshape = e.shape[1], w.shape
        return pred

    @staticmethod
    def backward(ctx, dpred):
        pooled, z, wv, *rest = ctx.saved_tensors
        dembed, dw = ops.duration_head_bwd(_f32c(dpred), z, pooled, rest[0] if rest else None, wv, ctx.T)
        return dembed, None, dw.view(ctx.wshape)


# ------------------------------------------------------------------------------------------------ MelSpec

def _melscale_fbanks_htk(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk'); built once on the host"""
    import math
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


class _Spectrogram(Module):
    def __init__(self, win_length):
        super().__init__()
        self.register_buffer('window', torch.hann_window(win_length, periodic=True))


class _MelScale(Module):
    def __init__(self, n_mels, sample_rate, n_stft):
        super().__init__()
        self.register_buffer('fb', _melscale_fbanks_htk(n_stft, 0., float(sample_rate // 2), n_mels, sample_rate))


class _MelSpectrogram(Module):      # buffer names of torchaudio.transforms.MelSpectrogram (state_dict keys, Appendix B)
    def __init__(self, sample_rate, n_fft, win_length, n_mels):
        super().__init__()
        self.spectrogram = _Spectrogram(win_length)
        self.mel_scale = _MelScale(n_mels, sample_rate, n_fft // 2 + 1)


class MelSpec(Module):
    """e2_tts.py:248-290: reflect-pad STFT magnitude -> htk mel filterbank -> log(clamp(1e-5)), on the HIP kernel."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=100,
                 sampling_rate=24_000, normalize=False, power=1, norm=None, center=True):
        super().__init__()
        if norm is not None or normalize or power != 1 or not center or win_length != filter_length:
            raise NotImplementedError('only the reference defaults (power=1, center, norm=None) are built')
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        self.filter_length, self.hop_length = filter_length, hop_length
        self.mel_stft = _MelSpectrogram(sampling_rate, filter_length, win_length, n_mel_channels)
        self.register_buffer('dummy', torch.tensor(0), persistent=False)

    def forward(self, inp, lens=None):
        """lens (extension, see data.py): samples per row of a zero-padded ragged batch.  Every row then comes out as the
        reference's one-clip-at-a-time transform followed by collate_fn's zero padding would give it (trainer.py:61-82)"""
        if inp.ndim == 3:
            inp = inp.squeeze(1)
        assert inp.ndim == 2
        if self.dummy.device != inp.device:
            self.to(inp.device)
        if inp.device.type == 'cpu' and not ops.host_ok():
            # CPU tensors (the trainer builds MelSpec inside HFDataset / E2Trainer on the host, trainer.py:96,122,188):
            # the same HIP kernel, with the wave staged to the current HIP device and the log-mel copied back.  Not a CPU
            # implementation -- without a HIP device this raises like every other op of the package.
            if not torch.cuda.is_available():
                raise ops.E2KError('MelSpec on a CPU tensor stages through the HIP device, and none is visible')
            dev = torch.device('cuda', torch.cuda.current_device())
            win, fb = self._staged_consts(dev)
            src = inp.float()
            src = src.pin_memory() if not src.is_pinned() else src
            out = ops.melspec(src.to(dev, non_blocking=True), win, fb, self.filter_length, self.hop_length,
                              lens=None if lens is None else lens.to(dev))
            return out.to('cpu')
        return ops.melspec(inp, self.mel_stft.spectrogram.window, self.mel_stft.mel_scale.fb,
                           self.filter_length, self.hop_length, lens=lens)          # (b, n_mels, frames)

    def _staged_consts(self, dev):
        c = self.__dict__.get('_staged')
        if c is None or c[0] != dev:
            c = (dev, self.mel_stft.spectrogram.window.to(dev), self.mel_stft.mel_scale.fb.to(dev))
            self.__dict__['_staged'] = c
        return c[1], c[2]

    def __deepcopy__(self, memo):       # (the staged device copies are a cache, not state: EMA deep-copies the model)
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != '_staged':
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new


# ------------------------------------------------------------------------------------------------ small modules

class SplitFreq(Module):                                       # Rearrange('b n (f d) -> b f n d') (e2_tts.py:1010,1208)
    def __init__(self, f):
        super().__init__()
        self.f = f

    def forward(self, x):
        b, n, fd = x.shape
        return x.reshape(b, n, self.f, fd // self.f).permute(0, 2, 1, 3)


class CharacterEmbed(Module):                                  # e2_tts.py:390-412
    def __init__(self, dim, num_embeds=256):
        super().__init__()
        self.dim = dim
        self.embed = nn.Embedding(num_embeds + 1, dim)

    def forward(self, text, max_seq_len, **kwargs):
        if _on_kernels(self.embed.weight) and text.dtype == torch.int64 and self.embed.weight.dtype == torch.float32 and self.dim % 4 == 0:
            return _CharEmbedFn.apply(text, self.embed.weight, max_seq_len)      # one gather kernel
        text = text + 1
        text = text[:, :max_seq_len]
        text = pad_to_length(text, max_seq_len, value=0)
        return self.embed(text)


class _AddLastDim(Module):                                      # stands where the reference has einops' Rearrange('... -> ... 1')
    def forward(self, x):
        return x[..., None]


class InterpolatedCharacterEmbed(Module):
    """e2_tts.py:414-484 (`interpolated_text=True`): each sample's character embeddings stretched linearly to its
    audio length plus an MLP of the fractional character position.  The reference loops over the batch with one
    `.item()` per sample; here the whole batch is one gather + lerp with no host synchronisation (front-end glue, not
    a kernel: B x T x dim_text elements)."""

    def __init__(self, dim, num_embeds=256):
        super().__init__()
        self.dim = dim
        self.embed = nn.Embedding(num_embeds, dim)
        self.abs_pos_mlp = nn.Sequential(_AddLastDim(), nn.Linear(1, dim), nn.SiLU(), nn.Linear(dim, dim))

    def _abs_pos(self, pos):
        """abs_pos_mlp (e2_tts.py:432-437): Linear(1, d) is an outer product (element-wise), the (B T, d) x (d, d) Linear behind the SiLU
        runs on the HIP GEMMs like every other Linear of the path (round 6: it went to the vendor BLAS through nn.Linear before)"""
        l1, l2 = self.abs_pos_mlp[1], self.abs_pos_mlp[3]
        if not (_on_kernels(pos) and self.dim % 8 == 0 and l2.weight.dtype == torch.float32):
            return self.abs_pos_mlp(pos)
        h = F.silu(pos[..., None] * l1.weight[:, 0] + l1.bias)
        return _OutProjFn.apply(h, l2.weight, l2.bias)

    def forward(self, text, max_seq_len, mask=None):
        B, dev = text.shape[0], text.device
        valid = text >= 0
        nt = valid.sum(dim=-1)                                                      # characters per sample
        # compact the valid tokens to the front (the tokenizer pads at the end, but a custom one may not)
        order = torch.argsort((~valid).to(torch.int8), dim=-1, stable=True)
        tok = text.gather(1, order).clamp(min=0)
        n_audio = mask.sum(dim=-1) if exists(mask) else torch.full((B,), max_seq_len, device=dev)
        i = torch.arange(max_seq_len, device=dev, dtype=torch.float32)[None, :]      # output position
        ntf, naf = nt.to(torch.float32)[:, None], n_audio.to(torch.float32)[:, None]
        # F.interpolate(mode='bilinear', align_corners=False) along the sequence: source coordinate of output i
        src = ((i + 0.5) * (ntf / naf.clamp(min=1.)) - 0.5).clamp(min=0.)
        i0 = src.floor()
        w = src - i0
        last = (nt - 1).clamp(min=0)[:, None]
        i0 = torch.minimum(i0.long(), last)
        i1 = torch.minimum(i0 + 1, last)
        emb = self.embed(tok)                                                       # (B, nt_max, d)
        e0 = emb.gather(1, i0[..., None].expand(-1, -1, self.dim))
        e1 = emb.gather(1, i1[..., None].expand(-1, -1, self.dim))
        out = e0 * (1. - w[..., None]) + e1 * w[..., None]
        inside = (i < naf) & (ntf > 0)                                                 # positions this sample has audio for
        # torch.linspace(0, nt, n_audio): step nt / (n_audio - 1); zero-padded beyond n_audio
        pos = torch.where(inside, i * (ntf / (naf - 1.).clamp(min=1.)), torch.zeros_like(i))
        out = torch.where(inside[..., None], out, torch.zeros_like(out)) + self._abs_pos(pos)
        if exists(mask):
            out = torch.where(mask[..., None], out, torch.zeros_like(out))
        return out


class HLGaussLoss(Module):
    """hl_gauss_pytorch.HLGaussLoss (e2_tts.py:966-967 hands its keyword dict through; SURVEY.md A.7, arXiv 2403.03950): a scalar
    target as the histogram of a Gaussian over `num_bins` bins of [min_value, max_value] (differences of erf at the bin edges,
    normalised by the mass inside the range), cross-entropy of the logits against it, prediction = softmax expectation of the bin
    centres.  (B, num_bins) element-wise work on the device the logits live on; `support` / `centers` are non-persistent buffers as in
    that package, so state_dicts carry neither."""

    def __init__(self, min_value, max_value, num_bins, sigma=None, sigma_to_bin_ratio=None, eps=1e-10, clamp_to_range=False):
        super().__init__()
        assert not (exists(sigma) and exists(sigma_to_bin_ratio))
        self.eps = eps
        support = torch.linspace(min_value, max_value, num_bins + 1).float()
        bin_size = (support[1] - support[0]).item()
        sigma = default(sigma, default(sigma_to_bin_ratio, 2.) * bin_size)
        assert sigma > 0.
        self.sigma, self.num_bins, self.min_value, self.max_value, self.clamp_to_range = sigma, num_bins, min_value, max_value, clamp_to_range
        self.register_buffer('support', support, persistent=False)
        self.register_buffer('centers', (support[:-1] + support[1:]) / 2, persistent=False)
        self.sigma_times_sqrt_two = 2. ** 0.5 * sigma

    def transform_from_logits(self, logits):
        return (logits.softmax(dim=-1) * self.centers).sum(dim=-1)

    def transform_to_probs(self, target):
        cdf = torch.special.erf((self.support - target[..., None]) / self.sigma_times_sqrt_two)
        z = cdf[..., -1:] - cdf[..., :1]
        return (cdf[..., 1:] - cdf[..., :-1]) / z.clamp(min=self.eps)

    def forward(self, logits, target=None):
        if not exists(target):
            return self.transform_from_logits(logits)
        if self.clamp_to_range:
            target = target.clamp(min=self.min_value, max=self.max_value)
        return F.cross_entropy(logits, self.transform_to_probs(target))


class HLGaussLayer(Module):                                    # hl_gauss_pytorch.HLGaussLayer as e2_tts.py:1035-1040 builds it (A.7)
    """regression (the reference's default): Linear(dim, 1, no bias) -> activation -> MSE; classification (`use_regression=False` with
    `hl_gauss_loss=dict(min_value, max_value, num_bins, ...)`, round 6): Linear(dim, num_bins, no bias) -> HLGaussLoss.  On the device
    the (B, dim) x (num_bins, dim) projection and its gradients run on the HIP GEMMs (_OutProjFn), like every other Linear of the path."""

    def __init__(self, dim, hl_gauss_loss=None, use_regression=True, regress_activation=None):
        super().__init__()
        if isinstance(hl_gauss_loss, dict):
            hl_gauss_loss = HLGaussLoss(**hl_gauss_loss)
        self.hl_gauss_loss = hl_gauss_loss
        self.use_classification = not use_regression
        assert not (self.use_classification and not exists(hl_gauss_loss)), '`hl_gauss_loss` is not defined, only regression is permitted'
        self.to_pred = nn.Linear(dim, hl_gauss_loss.num_bins if self.use_classification else 1, bias=False)
        self.act = default(regress_activation, nn.Identity())

    def forward(self, embed, target=None):
        if self.use_classification:
            w = self.to_pred.weight
            if _on_kernels(embed) and embed.ndim == 2 and w.shape[1] % 8 == 0:
                logits = _OutProjFn.apply(embed[:, None, :], w, torch.zeros(w.shape[0], dtype=torch.float32, device=w.device))[:, 0]
            else:
                logits = self.to_pred(embed)
            return self.hl_gauss_loss(logits, target)
        pred = self.act(self.to_pred(embed)).squeeze(-1)
        if not exists(target):
            return pred
        return F.mse_loss(pred, target)


def _check_token_ids(tok, num_embeds):
    """host-side token tensors straight from a tokenizer: nn.Embedding(num_embeds + 1) would raise on an id it has no row
    for (e2_tts.py:398,407-410); the gather kernel would only keep its read in bounds.  Free here (no device round trip)"""
    if tok.numel() and not tok.is_cuda:
        lo, hi = int(tok.min()), int(tok.max())
        if lo < -1 or hi >= num_embeds:
            raise IndexError(f'tokenizer produced ids in [{lo}, {hi}] but text_num_embeds = {num_embeds} (-1 = padding)')
    return tok


def _resolve_tokenizer(tokenizer, text_num_embeds):
    if callable(tokenizer):
        assert exists(text_num_embeds), '`text_num_embeds` must be given if supplying your own tokenizer encode function'
        return tokenizer, text_num_embeds
    if tokenizer == 'char_utf8':
        return list_str_to_tensor, 256
    if tokenizer == 'phoneme_en':
        raise NotImplementedError('phoneme_en needs g2p_en + nltk data (no network here); pass a callable tokenizer')
    raise ValueError(f'unknown tokenizer string {tokenizer}')


# ------------------------------------------------------------------------------------------------ DurationPredictor

class DurationPredictor(Module):                               # e2_tts.py:956-1113
    def __init__(self, transformer, num_channels=None, mel_spec_kwargs: dict = dict(), char_embed_kwargs: dict = dict(),
                 text_num_embeds=None, num_freq_tokens=1, hl_gauss_loss=None, use_regression=True,
                 tokenizer='char_utf8'):
        super().__init__()
        assert num_freq_tokens > 0
        self.num_freq_tokens, self.has_freq_axis = num_freq_tokens, num_freq_tokens > 1
        if isinstance(transformer, dict):
            transformer = dict(transformer)
            set_if_missing_key(transformer, 'has_freq_axis', self.has_freq_axis)
            transformer = Transformer(**transformer, cond_on_time=False)
        assert transformer.has_freq_axis == self.has_freq_axis
        self.mel_spec = MelSpec(**mel_spec_kwargs)
        self.num_channels = default(num_channels, self.mel_spec.n_mel_channels)
        self.transformer = transformer
        self.dim = transformer.dim
        if not self.has_freq_axis:                                  # e2_tts.py:1004-1011
            self.proj_in = nn.Linear(self.num_channels, self.dim)
        else:
            self.proj_in = nn.Sequential(nn.Linear(self.num_channels, self.dim * num_freq_tokens), SplitFreq(num_freq_tokens))
        self.tokenizer, text_num_embeds = _resolve_tokenizer(tokenizer, text_num_embeds)
        self.embed_text = CharacterEmbed(transformer.dim_text, num_embeds=text_num_embeds, **char_embed_kwargs)
        self.hl_gauss_layer = HLGaussLayer(self.dim, hl_gauss_loss=hl_gauss_loss, use_regression=use_regression,
                                           regress_activation=nn.Softplus())

    def forward(self, x, *, text=None, lens=None, return_loss=True, _rand_frac_index=None):
        if x.ndim == 2:
            x = self.mel_spec(x).transpose(1, 2)
            assert x.shape[-1] == self.dim                      # reference quirk (e2_tts.py:1055)
        x = self.proj_in(x)
        batch, seq_len, device = x.shape[0], x.shape[-2], x.device
        text_embed = None
        if exists(text):
            if isinstance(text, list):
                text = list_str_to_tensor(text).to(device)
                assert text.shape[0] == batch
            text_embed = self.embed_text(text, seq_len)
        if not exists(lens):
            lens = torch.full((batch,), seq_len, device=device)
        mask = lens_to_mask(lens, length=seq_len)
        if return_loss:
            rand_frac_index = _rand_frac_index if exists(_rand_frac_index) else x.new_zeros(batch).uniform_(0, 1)
            rand_index = (rand_frac_index * lens).long()
            seq = torch.arange(seq_len, device=device)
            mask = mask & (seq[None, :] < rand_index[:, None])
        embed = self.transformer(x, mask=mask, text_embed=text_embed)
        if self.has_freq_axis:                                      # e2_tts.py:1030,1098: mean over the frequency tokens
            embed = embed.mean(dim=1)
        hl = self.hl_gauss_layer
        if not hl.use_classification and _on_kernels(embed) and isinstance(hl.act, nn.Softplus) and hl.act.beta == 1 and hl.act.threshold == 20:
            # masked mean over the frames, the (dim -> 1) regression head and its Softplus in one kernel
            pred = _DurationHeadFn.apply(embed, mask, hl.to_pred.weight)
            return pred if not return_loss else F.mse_loss(pred, lens.float())
        pooled = maybe_masked_mean(embed, mask)
        if not return_loss:
            return hl(pooled)
        return hl(pooled, lens.float())


# ------------------------------------------------------------------------------------------------ E2TTS

def _odeint_midpoint(fn, y0, t):
    """torchdiffeq.odeint(method='midpoint') on the given grid; returns the final state only"""
    y = y0
    for t0, t1 in zip(t[:-1], t[1:]):
        dt = t1 - t0
        f0 = fn(t0, y)
        y_mid = y + f0 * (dt * 0.5)
        y = y + dt * fn(t0 + dt * 0.5, y_mid)
    return y


def _odeint_euler(fn, y0, t):
    """torchdiffeq.odeint(method='euler'): fixed grid, y += dt f(t0, y)"""
    y = y0
    for t0, t1 in zip(t[:-1], t[1:]):
        y = y + (t1 - t0) * fn(t0, y)
    return y


def _odeint_rk4(fn, y0, t):
    """torchdiffeq.odeint(method='rk4'): its fixed-grid fourth-order step is the 3/8 rule (`rk4_alt_step_func`)"""
    y = y0
    for t0, t1 in zip(t[:-1], t[1:]):
        dt = t1 - t0
        k1 = fn(t0, y)
        k2 = fn(t0 + dt / 3, y + dt * k1 / 3)
        k3 = fn(t0 + dt * 2 / 3, y + dt * (k2 - k1 / 3))
        k4 = fn(t1, y + dt * (k1 - k2 + k3))
        y = y + dt * (k1 + 3 * (k2 + k3) + k4) / 8
    return y


def _odeint_dopri5(fn, y0, t, rtol=1e-5, atol=1e-5):
    """torchdiffeq.odeint(method='dopri5'), the package's adaptive default (`odeint_kwargs` carries `atol` / `rtol` for it,
    e2_tts.py:1122-1126,1421): Dormand-Prince 5(4) with first-same-as-last, the mixed error norm
    rms(err / (atol + rtol max(|y0|, |y1|))), step factors clamped to [0.2, 10] with safety 0.9 and Hairer's starting step.
    torchdiffeq is not installed here, so this is a restatement of the published method, held to scipy's RK45 and to
    analytic solutions in tests/test_oracle.py (UNPINNED against the package).  Only the state at t[-1] is needed by
    sample(): the last step is clamped onto it (torchdiffeq steps past and interpolates; both are within the tolerance)."""
    c = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1., 1.)
    a = ((1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9), (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
         (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656), (35 / 384, 0., 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
    e = (35 / 384 - 5179 / 57600, 0., 500 / 1113 - 7571 / 16695, 125 / 192 - 393 / 640, -2187 / 6784 + 92097 / 339200, 11 / 84 - 187 / 2100, -1 / 40)

    def norm(x):
        return float(x.double().pow(2).mean().sqrt())
    t0, t1 = float(t[0]), float(t[-1])
    tt = lambda v: torch.as_tensor(v, dtype=t.dtype, device=t.device)
    y, f0 = y0, fn(tt(t0), y0)
    # starting step (Hairer, Norsett, Wanner I, II.4)
    sc = atol + rtol * y.abs()
    d0, d1 = norm(y / sc), norm(f0 / sc)
    h0 = 1e-6 if d0 < 1e-5 or d1 < 1e-5 else 0.01 * d0 / d1
    f1 = fn(tt(t0 + h0), y + h0 * f0)
    d2 = norm((f1 - f0) / sc) / h0
    h = min(100 * h0, max(1e-6, h0 * 1e-3) if max(d1, d2) <= 1e-15 else (0.01 / max(d1, d2)) ** (1 / 5))
    tc, k1, nfe = t0, f0, 2
    while tc < t1 - 1e-12:
        h = min(h, t1 - tc)
        ks = [k1]
        for ci, ai in zip(c, a):
            yi = y
            for aij, kj in zip(ai, ks):
                if aij:
                    yi = yi + (h * aij) * kj
            ks.append(fn(tt(tc + ci * h), yi))
        nfe += 6
        y_new = yi                                        # (the last stage is the 5th-order solution: first same as last)
        err = sum((h * ej) * kj for ej, kj in zip(e, ks) if ej)
        ratio = norm(err / (atol + rtol * torch.maximum(y.abs(), y_new.abs())))
        if ratio <= 1.:
            tc, y, k1 = tc + h, y_new, ks[-1]
        factor = 10. if ratio == 0. else min(10., max(0.2, 0.9 * ratio ** -0.2))
        h = h * factor
        assert nfe < 100000, 'dopri5: step size underflow'
    return y


def _adaptive(fn, y0, t, kw):
    # absent keys fall back to torchdiffeq.odeint's own defaults (rtol 1e-7, atol 1e-9), not to the reference constructor's
    return _odeint_dopri5(fn, y0, t, rtol=kw.get('rtol', 1e-7), atol=kw.get('atol', 1e-9))


def _method(kw):
    """an `odeint_kwargs` without `method` means torchdiffeq's default solver, dopri5 (the reference passes the dict through
    unchanged, e2_tts.py:1421; its own default dict names 'midpoint')"""
    return kw.get('method', 'dopri5')


_SOLVERS = {'midpoint': _odeint_midpoint, 'euler': _odeint_euler, 'rk4': _odeint_rk4, 'dopri5': _odeint_dopri5}


class E2TTS(Module):
    def __init__(
        self,
        transformer=None,
        duration_predictor=None,
        odeint_kwargs: dict = dict(atol=1e-5, rtol=1e-5, method='midpoint'),
        cond_drop_prob=0.25,
        num_channels=None,
        mel_spec_module=None,
        num_freq_tokens=1,
        char_embed_kwargs: dict = dict(),
        mel_spec_kwargs: dict = dict(),
        frac_lengths_mask=(0.7, 1.),
        concat_cond=False,
        interpolated_text=False,
        text_num_embeds=None,
        tokenizer='char_utf8',
        use_vocos=True,
        pretrained_vocos_path='charactr/vocos-mel-24khz',
        sampling_rate=None,
        velocity_consistency_weight=0.,
    ):
        super().__init__()
        assert num_freq_tokens > 0
        if _method(odeint_kwargs) not in _SOLVERS:
            raise NotImplementedError(f'solvers {sorted(_SOLVERS)} are built (the reference default is midpoint; dopri5 = torchdiffeq\'s adaptive default); '
                                      'adaptive torchdiffeq methods are not')
        self.num_freq_tokens, self.has_freq_axis = num_freq_tokens, num_freq_tokens > 1
        if isinstance(transformer, dict):
            transformer = dict(transformer)
            set_if_missing_key(transformer, 'has_freq_axis', self.has_freq_axis)
            transformer = Transformer(**transformer, cond_on_time=True)
        assert transformer.has_freq_axis == self.has_freq_axis
        self.transformer = transformer
        if isinstance(duration_predictor, dict):
            duration_predictor = DurationPredictor(**duration_predictor)
        dim, dim_text = transformer.dim, transformer.dim_text
        self.dim, self.dim_text = dim, dim_text
        self.frac_lengths_mask = frac_lengths_mask
        self.duration_predictor = duration_predictor
        self.odeint_kwargs = odeint_kwargs
        self._cfg_seen = {}        # (runtime state of _cfg_passes_concurrent: evaluations seen per signature)
        self.mel_spec = default(mel_spec_module, MelSpec(**mel_spec_kwargs))
        num_channels = default(num_channels, self.mel_spec.n_mel_channels)
        self.num_channels = num_channels
        self.sampling_rate = default(sampling_rate, getattr(self.mel_spec, 'sampling_rate', None))
        self.concat_cond = concat_cond                    # e2_tts.py:1196-1204: one projection of cat(cond, x) instead of two summed
        if concat_cond:
            self.proj_in = nn.Linear(num_channels * 2, dim * num_freq_tokens)
        else:
            self.proj_in = nn.Linear(num_channels, dim * num_freq_tokens)
            self.cond_proj_in = nn.Linear(num_channels, dim * num_freq_tokens)
        self.maybe_split_freq = SplitFreq(num_freq_tokens) if self.has_freq_axis else nn.Identity()      # e2_tts.py:1208-1210
        self.to_pred = nn.Linear(dim, num_channels)
        self.tokenizer, text_num_embeds = _resolve_tokenizer(tokenizer, text_num_embeds)
        self.cond_drop_prob = cond_drop_prob
        embed_klass = InterpolatedCharacterEmbed if interpolated_text else CharacterEmbed      # e2_tts.py:1236-1238
        self.embed_text = embed_klass(dim_text, num_embeds=text_num_embeds, **char_embed_kwargs)
        self.register_buffer('zero', torch.tensor(0.), persistent=False)
        self.velocity_consistency_weight = velocity_consistency_weight
        # the reference default downloads charactr/vocos-mel-24khz from the HF hub (e2_tts.py:1244); the vocoder is
        # outside the hot path and there is no network here
        self.vocos = None
        if use_vocos:
            try:
                from vocos import Vocos
                self.vocos = Vocos.from_pretrained(pretrained_vocos_path)
            except Exception as e:      # noqa: BLE001
                raise RuntimeError('use_vocos=True needs the `vocos` package and hub access; construct with '
                                   'use_vocos=False and pass `vocoder=` to sample()') from e

    @property
    def device(self):
        return next(self.parameters()).device

    def _num_text_ids(self):
        """ids a tokenizer may produce: CharacterEmbed shifts them by one into a table of text_num_embeds + 1 rows (row 0 = padding,
        e2_tts.py:398,407-410), InterpolatedCharacterEmbed indexes nn.Embedding(text_num_embeds) directly (e2_tts.py:429,468)"""
        rows = self.embed_text.embed.num_embeddings
        return rows if isinstance(self.embed_text, InterpolatedCharacterEmbed) else rows - 1

    def transformer_with_pred_head(self, x, cond, times, mask=None, text=None, drop_text_cond=None,
                                   return_drop_text_cond=False, _projected=None):
        seq_len = x.shape[-2]
        drop_text_cond = default(drop_text_cond, self.training and random() < self.cond_drop_prob)
        C = self.num_channels
        if not _on_kernels(x):
            raise ops.E2KError('e2_tts_pytorch_amd kernels need tensors on a HIP device (no CPU path)')
        if _projected is not None:                        # forward(): proj_in(x) + cond_proj_in(cond) already formed by _FlowInProjFn
            x = _projected
        elif self.concat_cond:                            # e2_tts.py:1263-1276: proj_in(cat(cond, x)), never concatenated
            w = self.proj_in.weight
            x = _InProjFn.apply(cond, x, w[:, :C], w[:, C:], self.proj_in.bias)
        else:                                             # e2_tts.py:1274-1277: proj_in(x) + cond_proj_in(cond), one GEMM
            x = _InProjFn.apply(x, cond, self.proj_in.weight, self.cond_proj_in.weight, self.proj_in.bias + self.cond_proj_in.bias)
        text_embed = None
        if exists(text) and not drop_text_cond:
            text_embed = self.embed_text(text, seq_len, mask=mask)
        # (the split of the frequency tokens commutes with the sum of the two projections, e2_tts.py:1268-1277)
        embed = self.transformer(self.maybe_split_freq(x), times=times, mask=mask, text_embed=text_embed)
        if self.has_freq_axis:                            # e2_tts.py:1210,1296: mean over the frequency tokens
            embed = embed.mean(dim=1)
        pred = _OutProjFn.apply(embed, self.to_pred.weight, self.to_pred.bias)
        if not return_drop_text_cond:
            return pred
        return pred, drop_text_cond

    def _cfg_passes_concurrent(self, args, kwargs, null_model, null_drop_text_cond):
        """sample(): the conditional and the null pass of one function evaluation (e2_tts.py:1303-1330) are independent until the CFG
        combine.  The null pass is issued on a second HIP stream, the conditional one on the caller's: the two passes' kernels share
        the chip -- the row kernels of one (HBM bound: hyper-connections, qkv_post, norms; 30 % of a pass) run under the GEMMs of the
        other (matrix bound), and each launch's partial last round of tiles is filled by the other stream.  Same kernels, same order
        inside each pass: bit-identical to the sequential schedule (tests/test_e2tts.py::test_cfg_passes_concurrent_bit_identical).

        What the passes share is read-only once it exists (bf16 weight shadows, rotary tables); things that are created on first use
        are created while the passes still run one after the other: the shadows are refreshed here, before the streams part, and the
        first two evaluations of a signature (eager pass, plan recording) are serialised by an extra join.  Launch plans are keyed by
        stream (backbone._plan_forward), so the null pass's plan and its split-K workspace belong to the side stream."""
        x = args[0]
        dev = x.device
        main = torch.cuda.current_stream(dev)
        side = _CFG_STREAMS.get(dev.index)
        if side is None:
            side = _CFG_STREAMS[dev.index] = torch.cuda.Stream(dev)
        for m in (self, null_model) if null_model is not self else (self,):
            sync = getattr(m.transformer, '_sync', None)
            if sync is not None:
                sync(dev)
            # the rotary table of this length as well: the backbone's cache is cleared once it holds more than 16 lengths, so a
            # length whose first two evaluations were serialised long ago may find its table gone -- and the null pass would then create
            # it on the side stream while the conditional pass reads it from the cache on the caller's, unordered
            rot = getattr(m.transformer, '_rot_table', None)
            if rot is not None:
                rot(x.shape[1] + m.transformer.num_registers, dev)
        key = (tuple(x.shape), exists(kwargs.get('text')), exists(kwargs.get('mask')), id(null_model), main.cuda_stream)
        seen = self._cfg_seen.get(key, 0)
        self._cfg_seen[key] = seen + 1
        if len(self._cfg_seen) > 256:
            self._cfg_seen = {key: seen + 1}
        side.wait_stream(main)
        with torch.cuda.stream(side):
            null_pred = null_model.transformer_with_pred_head(*args, drop_text_cond=null_drop_text_cond, **kwargs)
        if seen < 2:
            main.wait_stream(side)
        pred = self.transformer_with_pred_head(*args, drop_text_cond=False, **kwargs)
        main.wait_stream(side)
        null_pred.record_stream(main)          # (allocated on the side stream, consumed by the combine on the caller's)
        return pred, null_pred

    def cfg_transformer_with_pred_head(self, *args, cfg_strength: float = 1., cfg_null_model=None,
                                       remove_parallel_component: bool = True, keep_parallel_frac: float = 0., **kwargs):
        if cfg_strength < 1e-5:
            return self.transformer_with_pred_head(*args, drop_text_cond=False, **kwargs)
        null_drop_text_cond = not exists(cfg_null_model)
        cfg_null_model = default(cfg_null_model, self)
        if _CFG_CONCURRENT and not torch.is_grad_enabled() and args[0].is_cuda:
            pred, null_pred = self._cfg_passes_concurrent(args, kwargs, cfg_null_model, null_drop_text_cond)
        else:
            pred = self.transformer_with_pred_head(*args, drop_text_cond=False, **kwargs)
            null_pred = cfg_null_model.transformer_with_pred_head(*args, drop_text_cond=null_drop_text_cond, **kwargs)
        if not torch.is_grad_enabled() and pred.dtype == torch.float32 and _on_kernels(pred):
            # sample(): one kernel for the update, its fp64 projection and the combine
            return ops.cfg_combine(pred.contiguous(), null_pred.contiguous(), cfg_strength, keep_parallel_frac, remove_parallel_component)
        cfg_update = pred - null_pred
        if remove_parallel_component:
            parallel, orthogonal = project(cfg_update, pred)
            cfg_update = orthogonal + parallel * keep_parallel_frac
        return pred + cfg_update * cfg_strength

    @torch.no_grad()
    def sample(self, cond, *, text=None, lens=None, duration=None, steps=32, cfg_strength=1., cfg_null_model=None,
               max_duration=4096, vocoder=None, return_raw_output=None, save_to_filename=None, _y0=None):
        self.eval()
        if cond.ndim == 2:
            cond = self.mel_spec(cond).transpose(1, 2)
            assert cond.shape[-1] == self.num_channels
        batch, cond_seq_len, device = cond.shape[0], cond.shape[1], cond.device
        if not exists(lens):
            lens = torch.full((batch,), cond_seq_len, device=device, dtype=torch.long)
        if isinstance(text, list):
            text = _check_token_ids(self.tokenizer(text), self._num_text_ids()).to(device)
            assert text.shape[0] == batch
        if exists(text):
            text_lens = (text != -1).sum(dim=-1)
            lens = torch.maximum(text_lens, lens)
        cond_mask = lens_to_mask(lens)
        if exists(duration):
            if isinstance(duration, int):
                duration = torch.full((batch,), duration, device=device, dtype=torch.long)
        elif exists(self.duration_predictor):
            duration = self.duration_predictor(cond, text=text, lens=lens, return_loss=False).long()
        duration = torch.maximum(lens + 1, duration)
        duration = duration.clamp(max=max_duration)
        assert duration.shape[0] == batch
        max_dur = int(duration.amax())
        cond = F.pad(cond, (0, 0, 0, max_dur - cond_seq_len), value=0.)
        cond_mask = F.pad(cond_mask, (0, max_dur - cond_mask.shape[-1]), value=False)
        cond_mask = cond_mask[..., None]
        mask = lens_to_mask(duration)
        step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))

        def fn(t, x):
            return self.cfg_transformer_with_pred_head(x, step_cond, times=t, text=text, mask=mask,
                                                       cfg_strength=cfg_strength, cfg_null_model=cfg_null_model)

        y0 = _y0 if exists(_y0) else torch.randn_like(cond)
        t = torch.linspace(0, 1, steps, device=self.device)
        method = _method(self.odeint_kwargs)
        sampled = _adaptive(fn, y0, t, self.odeint_kwargs) if method == 'dopri5' else _SOLVERS[method](fn, y0, t)
        out = torch.where(cond_mask, cond, sampled)
        if exists(return_raw_output) and return_raw_output:
            return out
        if exists(vocoder):
            assert not exists(self.vocos), '`use_vocos` should not be turned on if you are passing in a custom `vocoder` on sampling'
            out = vocoder(out.transpose(1, 2))
        elif exists(self.vocos):
            raise NotImplementedError('vocos decode is outside the hot path (SURVEY.md section 2 row 3)')
        if exists(save_to_filename):
            raise NotImplementedError('audio file output needs torchaudio (not available); use the returned tensors')
        return out

    def forward(self, inp, *, text=None, times=None, lens=None, velocity_consistency_model=None,
                velocity_consistency_delta=1e-5, _noise=None):
        """`times` is accepted and ignored exactly like the reference (e2_tts.py:1473,1523).
        _noise: optional dict(x0, times, frac_lengths, span_rand, drop_text_cond) -- explicit draws for parity tests."""
        _noise = default(_noise, {})
        need_velocity_loss = exists(velocity_consistency_model) and self.velocity_consistency_weight > 0.     # e2_tts.py:1478
        if inp.ndim == 2:
            inp = self.mel_spec(inp).transpose(1, 2)
            assert inp.shape[-1] == self.num_channels
        batch, seq_len, dtype, device = inp.shape[0], inp.shape[1], inp.dtype, self.device
        if isinstance(text, list):
            text = _check_token_ids(self.tokenizer(text), self._num_text_ids())
            # pinned staging + non-blocking copy: a pageable H2D copy is a synchronising HIP call, i.e. the host would
            # wait here for the previous step's kernels before it can enqueue this one
            text = text.pin_memory().to(device, non_blocking=True) if device.type == 'cuda' else text.to(device)
            assert text.shape[0] == batch
        if not exists(lens):
            lens = torch.full((batch,), seq_len, device=device)
        mask = lens_to_mask(lens, length=seq_len)
        frac_lengths = _noise.get('frac_lengths')
        if frac_lengths is None:
            frac_lengths = torch.zeros((batch,), device=device).float().uniform_(*self.frac_lengths_mask)
        rand_span_mask = mask_from_frac_lengths(lens, frac_lengths, max_length=seq_len, rand=_noise.get('span_rand'))
        rand_span_mask = rand_span_mask & mask
        x1 = inp
        x0 = _noise['x0'] if 'x0' in _noise else torch.randn_like(x1)
        times = _noise['times'] if 'times' in _noise else torch.rand((batch,), dtype=dtype, device=device)
        t = times[:, None, None]
        if need_velocity_loss:                      # e2_tts.py:1528-1529: keep t + delta inside [0, 1]
            t = t * (1. - velocity_consistency_delta)
        fused = (_FUSE_FLOW_PROLOGUE and not need_velocity_loss and not self.concat_cond and _on_kernels(x1) and x1.dtype == torch.float32
                 and x0.dtype == torch.float32 and not (torch.is_grad_enabled() and (x1.requires_grad or x0.requires_grad or times.requires_grad)))
        if fused:
            # w, flow, cond and the input projection's operands in one kernel, the projection GEMM behind it
            h, flow, cond = _FlowInProjFn.apply(x0, x1, times, rand_span_mask, self.proj_in.weight, self.cond_proj_in.weight,
                                                self.proj_in.bias + self.cond_proj_in.bias)
            w = x1                       # (only its shape is read downstream)
        else:
            h = None
            w = (1. - t) * x0 + t * x1
            flow = x1 - x0
            cond = torch.where(rand_span_mask[..., None], torch.zeros_like(x1), x1)
        pred, did_drop = self.transformer_with_pred_head(w, cond, times=times, text=text, mask=mask,
                                                         drop_text_cond=_noise.get('drop_text_cond'),
                                                         return_drop_text_cond=True, _projected=h)
        m = rand_span_mask[..., None].to(pred.dtype)
        velocity_loss = self.zero
        if need_velocity_loss:
            # e2_tts.py:1558-1576: the EMA teacher (e.g. optim.FusedEMA(...).ema_model) predicts at t + delta with the
            # same text-drop decision, no gradient; masked-span mean of the squared difference (sync-free form)
            t_d = t + velocity_consistency_delta
            w_d = (1. - t_d) * x0 + t_d * x1
            with torch.no_grad():
                ema_pred = velocity_consistency_model.transformer_with_pred_head(
                    w_d, cond, times=times + velocity_consistency_delta, text=text, mask=mask, drop_text_cond=did_drop)
            velocity_loss = (F.mse_loss(pred, ema_pred, reduction='none') * m).sum() / (m.sum() * pred.shape[-1])
        # mean of the squared error over the masked span == loss[rand_span_mask].mean() (e2_tts.py:1580-1582), written
        # as a masked sum so that no boolean-index gather (device sync for the element count) is needed
        loss = _MaskedMSEFn.apply(pred, flow, rand_span_mask)
        total_loss = loss + velocity_loss * self.velocity_consistency_weight
        return E2TTSReturn(total_loss, cond, pred, x0 + pred, LossBreakdown(loss, velocity_loss))
