"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain-PyTorch (fp32 / fp64, CPU) restatement of the reference hot path
(lucidrains/e2-tts-pytorch @ 2025-03-21, /root/reference/e2_tts_pytorch/e2_tts.py)
plus the arithmetic of the third-party packages the reference imports but does
not vendor (x-transformers, hyper-connections, hl-gauss-pytorch, torchaudio,
torchdiffeq, einx -- SURVEY.md Appendix A).

PARITY: pinned to the reference's own source, unpinned for the third-party leaves.
oracle/pin_against_reference.py executes /root/reference/e2_tts_pytorch/e2_tts.py itself (with stand-ins, built from
the classes below, for the leaf modules of the packages that are not installed) and finds this file bit-identical to
it for Transformer forward + gradients, E2TTS.forward (incl. the CFG coin and velocity consistency), E2TTS.sample,
DurationPredictor, MelSpec and the helpers; tests/golden/reference_pinned.pt holds those reference outputs.  The
leaves themselves (Attention, FeedForward, RMSNorm, AdaptiveRMSNorm, RotaryEmbedding, HyperConnections, HLGaussLayer,
MelSpectrogram, midpoint odeint) restate the published algorithms and are checked only against (i) independent
implementations available here (torch.stft, transformers.audio_utils.mel_filter_bank,
F.scaled_dot_product_attention), (ii) analytic invariants of the reference's initialisation and (iii) fp64
self-consistency / finite differences -- see tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Module / parameter / buffer names equal the reference's
state_dict keys so weights can be exchanged 1:1.
"""
from __future__ import annotations

import math
import random as _pyrandom
from collections import namedtuple

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import Module, ModuleList

LossBreakdown = namedtuple('LossBreakdown', ['flow', 'velocity_consistency'])   # e2_tts.py:71
E2TTSReturn = namedtuple('E2TTS', ['loss', 'cond', 'pred_flow', 'pred_data', 'loss_breakdown'])  # e2_tts.py:73


def exists(v):
    return v is not None


def default(v, d):
    return v if exists(v) else d


# ---------------------------------------------------------------- helpers (e2_tts.py:113-235)

def project(x, y):
    """e2_tts.py:113-124 -- per-sample (flattened) projection of x on y, in fp64."""
    shape = x.shape
    dtype = x.dtype
    xf, yf = x.reshape(shape[0], -1).double(), y.reshape(shape[0], -1).double()
    unit = F.normalize(yf, dim=-1)
    parallel = (xf * unit).sum(dim=-1, keepdim=True) * unit
    orthogonal = xf - parallel
    return parallel.reshape(shape).to(dtype), orthogonal.reshape(shape).to(dtype)


def list_str_to_tensor(text, padding_value=-1):
    """e2_tts.py:128-135 -- UTF-8 byte tokenizer, pad -1 (the argument is ignored upstream too)."""
    ts = [torch.tensor([*bytes(t, 'UTF-8')], dtype=torch.long) for t in text]
    return nn.utils.rnn.pad_sequence(ts, padding_value=-1, batch_first=True)


def log_clamp(t, eps=1e-5):           # e2_tts.py:170
    return t.clamp(min=eps).log()


def lens_to_mask(t, length=None):     # e2_tts.py:173-182
    if not exists(length):
        length = int(t.amax())
    seq = torch.arange(length, device=t.device)
    return seq[None, :] < t[:, None]


def mask_from_start_end_indices(seq_len, start, end):   # e2_tts.py:184-191
    max_seq_len = int(seq_len.max().item())
    seq = torch.arange(max_seq_len, device=start.device).long()
    return (seq[None, :] >= start[:, None]) & (seq[None, :] < end[:, None])


def pad_to_length(t, length, value=None):   # e2_tts.py:226-235
    seq_len = t.shape[-1]
    if length > seq_len:
        t = F.pad(t, (0, length - seq_len), value=value)
    return t[..., :length]


def mask_from_frac_lengths(seq_len, frac_lengths, max_length=None, rand=None):   # e2_tts.py:193-210
    lengths = (frac_lengths * seq_len).long()
    max_start = seq_len - lengths
    if rand is None:
        rand = torch.rand_like(frac_lengths)
    start = (max_start * rand).long().clamp(min=0)
    end = start + lengths
    out = mask_from_start_end_indices(seq_len, start, end)
    if exists(max_length):
        out = pad_to_length(out, max_length)
    return out


def maybe_masked_mean(t, mask=None):   # e2_tts.py:212-224
    if not exists(mask):
        return t.mean(dim=1)
    t = torch.where(mask[..., None], t, torch.zeros_like(t))
    num = t.sum(dim=1)
    den = mask.float().sum(dim=1)
    return num / den.clamp(min=1.)[:, None]


# ---------------------------------------------------------------- MelSpec (e2_tts.py:248-290, torchaudio A.8)

def melscale_fbanks_htk(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') restated."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))       # (n_freqs, n_mels)


class _Spectrogram(Module):
    def __init__(self, n_fft, win_length, hop_length, power, center, normalized):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
        self.power, self.center, self.normalized = power, center, normalized
        self.register_buffer('window', torch.hann_window(win_length, periodic=True))

    def forward(self, x):
        spec = torch.stft(x, n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                          window=self.window, center=self.center, pad_mode='reflect',
                          normalized=False, onesided=True, return_complex=True)
        if self.normalized:
            spec = spec / self.window.pow(2.).sum().sqrt()
        mag = spec.abs()
        return mag if self.power == 1 else mag.pow(self.power)


class _MelScale(Module):
    def __init__(self, n_mels, sample_rate, n_stft):
        super().__init__()
        self.register_buffer('fb', melscale_fbanks_htk(n_stft, 0., float(sample_rate // 2), n_mels, sample_rate))

    def forward(self, spec):                       # (b, freq, time)
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)


class _MelSpectrogram(Module):
    def __init__(self, sample_rate, n_fft, win_length, hop_length, n_mels, power, center, normalized):
        super().__init__()
        self.spectrogram = _Spectrogram(n_fft, win_length, hop_length, power, center, normalized)
        self.mel_scale = _MelScale(n_mels, sample_rate, n_fft // 2 + 1)

    def forward(self, x):
        return self.mel_scale(self.spectrogram(x))


class MelSpec(Module):
    """e2_tts.py:248-290."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=100,
                 sampling_rate=24_000, normalize=False, power=1, norm=None, center=True):
        super().__init__()
        assert norm is None
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        self.mel_stft = _MelSpectrogram(sampling_rate, filter_length, win_length, hop_length,
                                        n_mel_channels, power, center, normalize)
        self.register_buffer('dummy', torch.tensor(0), persistent=False)

    def forward(self, inp):
        if inp.ndim == 3:
            inp = inp.squeeze(1)
        assert inp.ndim == 2
        if self.dummy.device != inp.device:
            self.to(inp.device)
        return log_clamp(self.mel_stft(inp))


# ---------------------------------------------------------------- torchaudio.transforms.Resample (trainer.py:116-118)

class Resample(Module):
    """torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults (resampling_method='sinc_interp_hann',
    lowpass_filter_width=6, rolloff=0.99), restated from the published torchaudio.functional._get_sinc_resample_kernel /
    _apply_sinc_resample_kernel (un-vendored, PARITY UNPINNED; tests/test_oracle.py checks it against an analytic band-limited
    signal): both rates are divided by their gcd; `new` filters of 2 width + orig taps -- a Hann-windowed sinc at
    rolloff x min(orig, new), one per output phase, built in fp64 and stored fp32 -- are applied as a conv1d of stride `orig` to
    the clip padded with `width` zeros in front and width + orig behind; the result is cut to ceil(new * length / orig) samples."""

    def __init__(self, orig_freq=16000, new_freq=16000, lowpass_filter_width=6, rolloff=0.99):
        super().__init__()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self.gcd = math.gcd(self.orig_freq, self.new_freq)
        self.lowpass_filter_width, self.rolloff = lowpass_filter_width, rolloff
        if self.orig_freq != self.new_freq:
            kernel, self.width = self.sinc_kernel(self.orig_freq, self.new_freq, self.gcd, lowpass_filter_width, rolloff)
            self.register_buffer('kernel', kernel, persistent=False)

    @staticmethod
    def sinc_kernel(orig_freq, new_freq, gcd, lowpass_filter_width=6, rolloff=0.99):
        orig, new = orig_freq // gcd, new_freq // gcd
        base = min(orig, new) * rolloff
        width = math.ceil(lowpass_filter_width * orig / base)
        idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
        t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
        t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        t = t * math.pi
        kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
        return kernels.float(), width                       # (new, 1, 2 width + orig)

    def forward(self, waveform):
        if self.orig_freq == self.new_freq:
            return waveform
        orig, new = self.orig_freq // self.gcd, self.new_freq // self.gcd
        shape = waveform.shape
        wav = waveform.reshape(-1, shape[-1])
        n, length = wav.shape
        wav = F.pad(wav, (self.width, self.width + orig))
        out = F.conv1d(wav[:, None], self.kernel, stride=orig).transpose(1, 2).reshape(n, -1)
        target = int(math.ceil(new * length / orig))
        return out[..., :target].reshape(shape[:-1] + (target,))


# ---------------------------------------------------------------- x-transformers pieces

class RMSNorm(Module):                       # A.1
    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.g = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return F.normalize(x, dim=-1) * self.scale * self.g


class AdaptiveRMSNorm(Module):               # A.2
    def __init__(self, dim, dim_condition=None):
        super().__init__()
        self.scale = dim ** 0.5
        self.to_gamma = nn.Linear(default(dim_condition, dim), dim, bias=False)
        nn.init.zeros_(self.to_gamma.weight)

    def forward(self, x, *, condition):
        if condition.ndim == 2:
            condition = condition[:, None, :]
        normed = F.normalize(x, dim=-1)
        gamma = self.to_gamma(condition)
        return normed * self.scale * (gamma + 1.)


class RotaryEmbedding(Module):               # A.6
    def __init__(self, dim, base=10000):
        super().__init__()
        self.register_buffer('inv_freq', 1. / (base ** (torch.arange(0, dim, 2).float() / dim)))

    def forward_from_seq_len(self, n):
        t = torch.arange(n, device=self.inv_freq.device).float()
        freqs = t[:, None] * self.inv_freq[None, :]
        freqs = torch.stack((freqs, freqs), dim=-1).flatten(-2)       # [t0,t0,t1,t1,...]
        return freqs[None], 1.


def rotate_half(x):
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def apply_rotary_pos_emb(t, freqs):
    freqs = freqs[:, -t.shape[-2]:, :]
    if t.ndim == 4 and freqs.ndim == 3:
        freqs = freqs[:, None]
    return t * freqs.cos() + rotate_half(t) * freqs.sin()


Intermediates = namedtuple('Intermediates', ['values'])


class Attention(Module):                     # A.3
    def __init__(self, dim, heads=8, dim_head=64, dropout=0., learned_value_residual_mix=False,
                 gate_value_heads=False, softclamp_logits=False, logit_softclamp_value=50.,
                 laser=False, laser_softclamp_value=15.):
        super().__init__()
        self.laser, self.laser_softclamp_value = laser, laser_softclamp_value
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        inner = heads * dim_head
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)
        self.to_v_head_gate = None
        if gate_value_heads:
            self.to_v_head_gate = nn.Linear(dim, heads)
            nn.init.constant_(self.to_v_head_gate.weight, 0)
            nn.init.constant_(self.to_v_head_gate.bias, 10)
        self.to_value_residual_mix = None
        if learned_value_residual_mix:
            self.to_value_residual_mix = nn.Sequential(nn.Linear(dim, heads), nn.Sigmoid())
        self.softclamp_logits = softclamp_logits
        self.logit_softclamp_value = logit_softclamp_value
        self.dropout_p = dropout
        self.dropout_mask = None      # test hook: explicit keep-mask (b,h,i,j) already scaled by 1/(1-p)

    def forward(self, x, mask=None, rotary_pos_emb=None, value_residual=None, return_intermediates=False):
        b, n, h = x.shape[0], x.shape[1], self.heads
        q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)
        q, k, v = (t.view(b, n, h, -1).transpose(1, 2) for t in (q, k, v))
        orig_values = v
        if exists(value_residual):
            # learned per-head mix, or the constant 0.5 when `learned_value_residual_mix` is off (the frequency attention of
            # `has_freq_axis`, e2_tts.py:655,925: x-transformers' `always(0.5)`)
            mix = self.to_value_residual_mix(x).transpose(1, 2)[..., None] if exists(self.to_value_residual_mix) else 0.5
            v = value_residual.lerp(v, mix)
        if exists(rotary_pos_emb):
            freqs, _ = rotary_pos_emb
            q = apply_rotary_pos_emb(q, freqs)
            k = apply_rotary_pos_emb(k, freqs)
        if self.laser:
            # LASER attention (e2_tts.py:543-544,641; x-transformers `laser`): attend over exp(softclamp(v)), take the log of the
            # result (clamped at 1e-20) before the head gates.  UNPINNED restatement of the third-party module.
            c = self.laser_softclamp_value
            v = ((v / c).tanh() * c).exp()
        sim = torch.einsum('bhid,bhjd->bhij', q, k) * self.scale
        if self.softclamp_logits:
            sim = (sim / self.logit_softclamp_value).tanh() * self.logit_softclamp_value
        if exists(mask):
            sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(sim.dtype).max)
        attn = sim.softmax(dim=-1, dtype=torch.float32).to(sim.dtype)
        if self.dropout_mask is not None:
            attn = attn * self.dropout_mask
        elif self.training and self.dropout_p > 0:
            attn = F.dropout(attn, self.dropout_p)
        out = torch.einsum('bhij,bhjd->bhid', attn, v)
        if self.laser:
            out = out.clamp(min=1e-20).log()
        if exists(self.to_v_head_gate):
            gate = self.to_v_head_gate(x).sigmoid()                              # b n h
            out = out * gate.transpose(1, 2)[..., None]
        out = out.transpose(1, 2).reshape(b, n, -1)
        out = self.to_out(out)
        if exists(mask):
            out = torch.where(mask[..., None], out, torch.zeros_like(out))
        if return_intermediates:
            return out, Intermediates(orig_values)
        return out


class _GLU(Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class _Dropout(Module):
    def __init__(self, p):
        super().__init__()
        self.p = p
        self.mask = None              # test hook: explicit keep-mask already scaled by 1/(1-p)

    def forward(self, x):
        if self.mask is not None:
            return x * self.mask
        return F.dropout(x, self.p, self.training)


class FeedForward(Module):                   # A.4
    def __init__(self, dim, glu=True, mult=4, dropout=0.):
        super().__init__()
        assert glu
        inner = int(dim * mult)
        self.ff = nn.Sequential(_GLU(dim, inner), _Dropout(dropout), nn.Linear(inner, dim))

    def forward(self, x):
        return self.ff(x)


# ---------------------------------------------------------------- hyper-connections

class _HCNorm(Module):
    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        return F.normalize(x, dim=-1) * self.scale * (self.gamma + 1)


class HyperConnections(Module):
    def __init__(self, num_residual_streams, *, dim, layer_index=None):
        super().__init__()
        s = num_residual_streams
        self.s = s
        self.norm = _HCNorm(dim)
        init_idx = default(layer_index, _pyrandom.randrange(s)) % s
        self.static_beta = nn.Parameter(torch.ones(s))
        a0 = torch.zeros(s, 1)
        a0[init_idx, 0] = 1.
        self.static_alpha = nn.Parameter(torch.cat([a0, torch.eye(s)], dim=1))
        self.dynamic_alpha_fn = nn.Parameter(torch.zeros(dim, s + 1))
        self.dynamic_alpha_scale = nn.Parameter(torch.ones(()) * 1e-2)
        self.dynamic_beta_fn = nn.Parameter(torch.zeros(dim))
        self.dynamic_beta_scale = nn.Parameter(torch.ones(()) * 1e-2)

    def forward(self, residuals):
        s = self.s
        bs, n, d = residuals.shape
        r = residuals.view(bs // s, s, n, d).permute(0, 2, 1, 3)                 # '(b s) n d -> b n s d'
        z = self.norm(r)
        alpha = (z @ self.dynamic_alpha_fn).tanh() * self.dynamic_alpha_scale + self.static_alpha   # b n s s+1
        beta = (z @ self.dynamic_beta_fn).tanh() * self.dynamic_beta_scale + self.static_beta       # b n s
        mix = torch.einsum('bnst,bnsd->bntd', alpha, r)
        branch_input, resid = mix[..., 0, :], mix[..., 1:, :]

        def add_residual(y):
            out = y[..., None, :] * beta[..., None] + resid                     # b n s d
            return out.permute(0, 2, 1, 3).reshape(bs, n, d)

        return branch_input, add_residual


def hc_expand(x, s):
    return x.repeat_interleave(s, dim=0)        # 'b ... -> (b s) ...'


def hc_reduce(x, s):
    return x.view(x.shape[0] // s, s, *x.shape[1:]).sum(dim=1)


# ---------------------------------------------------------------- local blocks (e2_tts.py:295-513)

class DepthwiseConv(Module):                 # e2_tts.py:295-328
    def __init__(self, dim, *, kernel_size):
        super().__init__()
        assert kernel_size % 2 == 1
        self.dw_conv1d = nn.Sequential(nn.Conv1d(dim, dim, kernel_size, groups=dim, padding=kernel_size // 2), nn.SiLU())

    def forward(self, x, mask=None):
        if exists(mask):
            x = torch.where(mask[..., None], x, torch.zeros_like(x))
        out = self.dw_conv1d(x.transpose(1, 2)).transpose(1, 2)
        if exists(mask):
            out = torch.where(mask[..., None], out, torch.zeros_like(out))
        return out


class AdaLNZero(Module):                     # e2_tts.py:332-351
    def __init__(self, dim, dim_condition=None, init_bias_value=-2.):
        super().__init__()
        self.to_gamma = nn.Linear(default(dim_condition, dim), dim)
        nn.init.zeros_(self.to_gamma.weight)
        nn.init.constant_(self.to_gamma.bias, init_bias_value)

    def forward(self, x, *, condition):
        if condition.ndim == 2:
            condition = condition[:, None, :]
        return x * self.to_gamma(condition).sigmoid()


class Identity(Module):                      # e2_tts.py:107
    def forward(self, x, **kwargs):
        return x


class RandomFourierEmbed(Module):            # e2_tts.py:355-364
    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0
        self.register_buffer('weights', torch.randn(dim // 2))

    def forward(self, x):
        freqs = x[:, None] * self.weights[None, :] * 2 * torch.pi
        return torch.cat((x[:, None], freqs.sin(), freqs.cos()), dim=-1)


class CharacterEmbed(Module):                # e2_tts.py:390-412
    def __init__(self, dim, num_embeds=256):
        super().__init__()
        self.dim = dim
        self.embed = nn.Embedding(num_embeds + 1, dim)

    def forward(self, text, max_seq_len, **kwargs):
        text = text + 1
        text = text[:, :max_seq_len]
        text = pad_to_length(text, max_seq_len, value=0)
        return self.embed(text)


class _AddLastDim(Module):                   # einops Rearrange('... -> ... 1') (keeps the Sequential indices of the state_dict)
    def forward(self, x):
        return x[..., None]


class InterpolatedCharacterEmbed(Module):    # e2_tts.py:414-484
    """every sample's character embeddings are stretched (bilinear, i.e. linear along the sequence) to that sample's
    audio length; an MLP of the fractional character position is added"""

    def __init__(self, dim, num_embeds=256):
        super().__init__()
        self.dim = dim
        self.embed = nn.Embedding(num_embeds, dim)
        self.abs_pos_mlp = nn.Sequential(_AddLastDim(), nn.Linear(1, dim), nn.SiLU(), nn.Linear(dim, dim))

    def forward(self, text, max_seq_len, mask=None):
        embeds, positions = [], []
        for b in range(text.shape[0]):
            one = text[b][text[b] >= 0]
            e = self.embed(one)                                                    # (nt, d)
            nt = one.shape[0]
            n_audio = int(mask[b].sum().item()) if exists(mask) else max_seq_len
            e = F.interpolate(e.t()[None, :, :, None], (n_audio, 1), mode='bilinear')[0, :, :, 0].t()
            embeds.append(e)
            positions.append(torch.linspace(0, nt, n_audio, device=text.device))
        embeds = nn.utils.rnn.pad_sequence(embeds, batch_first=True)
        positions = nn.utils.rnn.pad_sequence(positions, batch_first=True)
        embeds = F.pad(embeds, (0, 0, 0, max_seq_len - embeds.shape[-2]))
        positions = pad_to_length(positions, max_seq_len)
        embeds = embeds + self.abs_pos_mlp(positions)
        if exists(mask):
            embeds = torch.where(mask[..., None], embeds, torch.zeros_like(embeds))
        return embeds


class TextAudioCrossCondition(Module):       # e2_tts.py:486-513
    def __init__(self, dim, dim_text, cond_audio_to_text=True):
        super().__init__()
        self.text_to_audio = nn.Linear(dim_text + dim, dim, bias=False)
        nn.init.zeros_(self.text_to_audio.weight)
        self.cond_audio_to_text = cond_audio_to_text
        if cond_audio_to_text:
            self.audio_to_text = nn.Linear(dim + dim_text, dim_text, bias=False)
            nn.init.zeros_(self.audio_to_text.weight)

    def forward(self, audio, text):
        audio_text = torch.cat((audio, text), dim=-1)
        text_cond = self.text_to_audio(audio_text)
        audio_cond = self.audio_to_text(audio_text) if self.cond_audio_to_text else 0.
        return audio + text_cond, text + audio_cond


class LinearFourierEmbed(Module):
    """e2_tts.py:368-386 -- bias-free linear whose first int(p dim) outputs pass through sin and cos"""
    def __init__(self, dim, p=0.5):
        super().__init__()
        assert p <= 1.
        dim_fourier = int(p * dim)
        dim_rest = dim - dim_fourier * 2
        self.linear = nn.Linear(dim, dim_fourier + dim_rest, bias=False)
        self.split_dims = (dim_fourier, dim_rest)

    def forward(self, x):
        fourier, rest = self.linear(x).split(self.split_dims, dim=-1)
        return torch.cat((fourier.sin(), fourier.cos(), rest), dim=-1)


# ---------------------------------------------------------------- Transformer (e2_tts.py:518-952)

class Transformer(Module):
    def __init__(self, *, dim, dim_text=None, depth=8, heads=8, dim_head=64, ff_mult=4, text_depth=None,
                 text_heads=None, text_dim_head=None, text_ff_mult=None, has_freq_axis=False, freq_heads=None,
                 freq_dim_head=None, cond_on_time=True, abs_pos_emb=True, max_seq_len=8192, kernel_size=31,
                 dropout=0.1, num_registers=32, scale_residual=False, attn_laser=False,
                 attn_laser_softclamp_value=15., attn_fourier_embed_input=False,
                 attn_fourier_embed_input_frac=0.25, num_residual_streams=4,
                 attn_kwargs=dict(gate_value_heads=True, softclamp_logits=True), ff_kwargs=dict()):
        super().__init__()
        assert depth % 2 == 0, 'depth needs to be even'
        assert num_residual_streams > 1
        self.max_seq_len = max_seq_len
        self.abs_pos_emb = nn.Embedding(max_seq_len, dim) if abs_pos_emb else None
        self.dim = dim
        dim_text = default(dim_text, dim // 2)
        self.dim_text = dim_text
        text_heads = default(text_heads, heads)
        text_dim_head = default(text_dim_head, dim_head)
        text_ff_mult = default(text_ff_mult, ff_mult)
        text_depth = default(text_depth, depth)
        assert 1 <= text_depth <= depth
        freq_heads = default(freq_heads, heads)                    # e2_tts.py:577-578
        freq_dim_head = default(freq_dim_head, dim_head)
        self.has_freq_axis = has_freq_axis
        self.depth = depth
        self.num_registers = num_registers
        self.registers = nn.Parameter(torch.zeros(num_registers, dim))
        nn.init.normal_(self.registers, std=0.02)
        self.text_registers = nn.Parameter(torch.zeros(num_registers, dim_text))
        nn.init.normal_(self.text_registers, std=0.02)
        self.rotary_emb = RotaryEmbedding(dim_head)
        self.text_rotary_emb = RotaryEmbedding(text_dim_head)
        if has_freq_axis:
            self.freq_rotary_emb = RotaryEmbedding(freq_dim_head)
        self.num_residual_streams = s = num_residual_streams
        self.cond_on_time = cond_on_time
        norm_klass = (lambda: AdaptiveRMSNorm(dim)) if cond_on_time else (lambda: RMSNorm(dim))
        post_klass = (lambda: AdaLNZero(dim)) if cond_on_time else Identity
        self.time_cond_mlp = Identity()
        if cond_on_time:
            self.time_cond_mlp = nn.Sequential(RandomFourierEmbed(dim), nn.Linear(dim + 1, dim), nn.SiLU())
        layers, hyper_conns = [], []
        for ind in range(depth):
            first = ind == 0
            later_half = ind >= depth // 2
            has_text = ind < text_depth
            laser = dict(laser=attn_laser, laser_softclamp_value=attn_laser_softclamp_value)
            freq_norm = freq_attn = freq_adaln = None
            if has_freq_axis:                                      # e2_tts.py:653-656: a default-keyword Attention (no gates, no soft-clamp)
                freq_norm = norm_klass()
                freq_attn = Attention(dim=dim, heads=freq_heads, dim_head=freq_dim_head)
                freq_adaln = post_klass()
            speech_modules = ModuleList([
                nn.Linear(dim * 2, dim, bias=False) if later_half else None,
                DepthwiseConv(dim, kernel_size=kernel_size),
                norm_klass(),
                Attention(dim=dim, heads=heads, dim_head=dim_head, dropout=dropout,
                          learned_value_residual_mix=not first, **laser, **attn_kwargs),
                LinearFourierEmbed(dim, p=attn_fourier_embed_input_frac) if attn_fourier_embed_input else nn.Identity(),
                post_klass(),
                norm_klass(),
                FeedForward(dim=dim, glu=True, mult=ff_mult, dropout=dropout, **ff_kwargs),
                post_klass(),
                freq_norm, freq_attn, freq_adaln])
            speech_hc = ModuleList([HyperConnections(s, dim=dim) for _ in range(3)] +
                                   [HyperConnections(s, dim=dim) if has_freq_axis else None])
            text_modules = text_hc = None
            if has_text:
                text_modules = ModuleList([
                    DepthwiseConv(dim_text, kernel_size=kernel_size),
                    RMSNorm(dim_text),
                    Attention(dim=dim_text, heads=text_heads, dim_head=text_dim_head, dropout=dropout,
                              learned_value_residual_mix=not first, **laser, **attn_kwargs),
                    RMSNorm(dim_text),
                    FeedForward(dim=dim_text, glu=True, mult=text_ff_mult, dropout=dropout, **ff_kwargs),
                    TextAudioCrossCondition(dim=dim, dim_text=dim_text, cond_audio_to_text=ind != text_depth - 1)])
                text_hc = ModuleList([HyperConnections(s, dim=dim_text) for _ in range(3)])
            hyper_conns.append(ModuleList([speech_hc, text_hc]))
            layers.append(ModuleList([speech_modules, text_modules]))
        self.layers = ModuleList(layers)
        self.hyper_conns = ModuleList(hyper_conns)
        self.final_norm = RMSNorm(dim)

    def forward(self, x, times=None, mask=None, text_embed=None):
        orig_batch = x.shape[0]
        assert (x.ndim == 4) == self.has_freq_axis
        freq_seq_len = 1
        if self.has_freq_axis:                                    # e2_tts.py:744-752: frequency tokens ride in the batch
            freq_seq_len = x.shape[1]
            x = x.reshape(orig_batch * freq_seq_len, *x.shape[2:])
            if exists(text_embed):
                text_embed = text_embed.repeat_interleave(freq_seq_len, dim=0)
            if exists(mask):
                mask = mask.repeat_interleave(freq_seq_len, dim=0)
        batch, seq_len, device = x.shape[0], x.shape[1], x.device
        assert not (exists(times) ^ self.cond_on_time)
        s = self.num_residual_streams
        if exists(self.abs_pos_emb):
            assert seq_len <= self.max_seq_len
            x = x + self.abs_pos_emb(torch.arange(seq_len, device=device))
        x = torch.cat((self.registers[None].expand(batch, -1, -1), x), dim=1)
        if exists(mask):
            mask = F.pad(mask, (self.num_registers, 0), value=True)
        norm_kwargs, freq_norm_kwargs = dict(), dict()
        if exists(times):
            if times.ndim == 0:
                times = times[None].expand(orig_batch)
            times = self.time_cond_mlp(times)
            if self.has_freq_axis:                                # e2_tts.py:784-789
                freq_norm_kwargs.update(condition=times.repeat_interleave(x.shape[-2], dim=0))
            times = times.repeat_interleave(freq_seq_len, dim=0)
            norm_kwargs.update(condition=times)
        rotary_pos_emb = self.rotary_emb.forward_from_seq_len(x.shape[-2])
        if self.has_freq_axis:
            freq_rotary_pos_emb = self.freq_rotary_emb.forward_from_seq_len(freq_seq_len)
        if exists(text_embed):
            text_rotary_pos_emb = self.text_rotary_emb.forward_from_seq_len(x.shape[-2])
            text_embed = torch.cat((self.text_registers[None].expand(batch, -1, -1), text_embed), dim=1)
        skips = []
        text_attn_first_values = None
        freq_attn_first_values = None
        attn_first_values = None
        x = hc_expand(x, s)
        if exists(text_embed):
            text_embed = hc_expand(text_embed, s)
        for ind, ((speech_modules, text_modules), (speech_hc, text_hc)) in enumerate(zip(self.layers, self.hyper_conns)):
            layer = ind + 1
            (skip_proj, speech_conv, attn_norm, attn, attn_input_fourier_embed, attn_adaln, ff_norm, ff, ff_adaln,
             freq_attn_norm, freq_attn, freq_attn_adaln) = speech_modules
            conv_residual, attn_residual, ff_residual, freq_attn_residual = speech_hc
            if exists(text_embed) and exists(text_modules):
                text_conv, text_attn_norm, text_attn, text_ff_norm, text_ff, cross_condition = text_modules
                t_conv_res, t_attn_res, t_ff_res = text_hc
                text_embed, add_residual = t_conv_res(text_embed)
                text_embed = text_conv(text_embed, mask=mask)
                text_embed = add_residual(text_embed)
                text_embed, add_residual = t_attn_res(text_embed)
                text_attn_out, inter = text_attn(text_attn_norm(text_embed), rotary_pos_emb=text_rotary_pos_emb,
                                                 mask=mask, return_intermediates=True,
                                                 value_residual=text_attn_first_values)
                text_embed = add_residual(text_attn_out)
                text_attn_first_values = default(text_attn_first_values, inter.values)
                text_embed, add_residual = t_ff_res(text_embed)
                text_embed = text_ff(text_ff_norm(text_embed))
                text_embed = add_residual(text_embed)
                x, text_embed = cross_condition(x, text_embed)
            if layer <= self.depth // 2:
                skips.append(x)
            else:
                skip = skips.pop()
                x = torch.cat((x, skip), dim=-1)
                x = skip_proj(x)
            x, add_residual = conv_residual(x)
            x = speech_conv(x, mask=mask)
            x = add_residual(x)
            x, add_residual = attn_residual(x)
            x = attn_norm(x, **norm_kwargs)
            x = attn_input_fourier_embed(x)
            attn_out, inter = attn(x, rotary_pos_emb=rotary_pos_emb, mask=mask, return_intermediates=True,
                                   value_residual=attn_first_values)
            attn_out = attn_adaln(attn_out, **norm_kwargs)
            x = add_residual(attn_out)
            attn_first_values = default(attn_first_values, inter.values)
            if self.has_freq_axis:                                # e2_tts.py:920-932: attention across the frequency tokens of a frame
                x, add_residual = freq_attn_residual(x)
                f, n, d = freq_seq_len, x.shape[-2], x.shape[-1]
                x = x.reshape(orig_batch, f, n, d).transpose(1, 2).reshape(orig_batch * n, f, d)
                attn_out, inter = freq_attn(freq_attn_norm(x, **freq_norm_kwargs), rotary_pos_emb=freq_rotary_pos_emb,
                                            return_intermediates=True, value_residual=freq_attn_first_values)
                attn_out = freq_attn_adaln(attn_out, **freq_norm_kwargs)
                attn_out = attn_out.reshape(orig_batch, n, f, d).transpose(1, 2).reshape(orig_batch * f, n, d)
                x = add_residual(attn_out)
                freq_attn_first_values = default(freq_attn_first_values, inter.values)
            x, add_residual = ff_residual(x)
            ff_out = ff(ff_norm(x, **norm_kwargs))
            ff_out = ff_adaln(ff_out, **norm_kwargs)
            x = add_residual(ff_out)
        assert len(skips) == 0
        x = x[:, self.num_registers:]
        x = hc_reduce(x, s)
        if self.has_freq_axis:
            x = x.reshape(orig_batch, freq_seq_len, *x.shape[1:])
        return self.final_norm(x)


# ---------------------------------------------------------------- HLGaussLayer, both modes

class HLGaussLoss(Module):
    """hl_gauss_pytorch.HLGaussLoss(min_value, max_value, num_bins, sigma=None, sigma_to_bin_ratio=None, eps=1e-10,
    clamp_to_range=False) -- "Stop regressing" (arXiv 2403.03950), restated from the published package (un-vendored, PARITY
    UNPINNED like the other third-party leaves): `support` = linspace(min, max, num_bins + 1), `centers` its midpoints (both
    non-persistent buffers), sigma = sigma_to_bin_ratio (default 2) x bin width.  A scalar target becomes the histogram of a
    Gaussian around it: differences of erf((support - y) / (sqrt(2) sigma)) between consecutive bin edges, normalised by the mass
    inside [min, max]; the loss is the cross-entropy of the logits against that histogram; a prediction is the softmax
    expectation of the bin centres."""

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
        self.sigma_times_sqrt_two = math.sqrt(2.) * sigma

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


class HLGaussLayer(Module):
    """hl_gauss_pytorch.HLGaussLayer(dim, hl_gauss_loss=dict-or-HLGaussLoss-or-None, use_regression, regress_activation) as the
    reference builds it (e2_tts.py:1035-1040; norm_embed left False): regression = Linear(dim, 1, no bias) -> activation -> MSE;
    classification (use_regression=False, needs hl_gauss_loss) = Linear(dim, num_bins, no bias) -> HLGaussLoss."""

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
            return self.hl_gauss_loss(self.to_pred(embed), target)
        pred = self.act(self.to_pred(embed)).squeeze(-1)
        if not exists(target):
            return pred
        return F.mse_loss(pred, target)


class SplitFreq(Module):
    """Rearrange('b n (f d) -> b f n d') (e2_tts.py:1010,1208): the projection's channels hold the frequency tokens f-major"""
    def __init__(self, f):
        super().__init__()
        self.f = f

    def forward(self, x):
        b, n, fd = x.shape
        return x.reshape(b, n, self.f, fd // self.f).permute(0, 2, 1, 3)


# ---------------------------------------------------------------- DurationPredictor (e2_tts.py:956-1113)

class DurationPredictor(Module):
    def __init__(self, transformer, num_channels=None, mel_spec_kwargs=dict(), char_embed_kwargs=dict(),
                 text_num_embeds=None, num_freq_tokens=1, hl_gauss_loss=None, use_regression=True,
                 tokenizer='char_utf8'):
        super().__init__()
        assert num_freq_tokens > 0
        self.num_freq_tokens, self.has_freq_axis = num_freq_tokens, num_freq_tokens > 1       # e2_tts.py:977-989
        if isinstance(transformer, dict):
            transformer = dict(transformer)
            transformer.setdefault('has_freq_axis', self.has_freq_axis)
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
        if callable(tokenizer):
            assert exists(text_num_embeds)
            self.tokenizer = tokenizer
        elif tokenizer == 'char_utf8':
            text_num_embeds = 256
            self.tokenizer = list_str_to_tensor
        else:
            raise ValueError(f'unknown tokenizer string {tokenizer}')
        self.embed_text = CharacterEmbed(transformer.dim_text, num_embeds=text_num_embeds, **char_embed_kwargs)
        self.hl_gauss_layer = HLGaussLayer(self.dim, hl_gauss_loss=hl_gauss_loss, use_regression=use_regression,
                                           regress_activation=nn.Softplus())

    def forward(self, x, *, text=None, lens=None, return_loss=True, _rand_frac_index=None):
        if x.ndim == 2:
            x = self.mel_spec(x).transpose(1, 2)
            assert x.shape[-1] == self.dim                       # reference quirk, e2_tts.py:1055
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
        if self.has_freq_axis:                                      # e2_tts.py:1030,1098: Reduce('b f n d -> b n d', 'mean')
            embed = embed.mean(dim=1)
        pooled = maybe_masked_mean(embed, mask)
        if not return_loss:
            return self.hl_gauss_layer(pooled)
        return self.hl_gauss_layer(pooled, lens.float())


# ---------------------------------------------------------------- midpoint ODE (torchdiffeq)

def odeint_midpoint(fn, y0, t):
    ys = [y0]
    y = y0
    for t0, t1 in zip(t[:-1], t[1:]):
        dt = t1 - t0
        f0 = fn(t0, y)
        y_mid = y + f0 * (dt * 0.5)
        y = y + dt * fn(t0 + dt * 0.5, y_mid)
        ys.append(y)
    return torch.stack(ys)


def odeint_fixed(fn, y0, t, method='midpoint'):
    """the other fixed-grid torchdiffeq solvers (restated, unpinned): 'euler', and 'rk4' = the 3/8-rule step
    torchdiffeq uses on a fixed grid (rk4_alt_step_func)"""
    if method == 'midpoint':
        return odeint_midpoint(fn, y0, t)
    ys, y = [y0], y0
    for t0, t1 in zip(t[:-1], t[1:]):
        dt = t1 - t0
        if method == 'euler':
            y = y + dt * fn(t0, y)
        elif method == 'rk4':
            k1 = fn(t0, y)
            k2 = fn(t0 + dt / 3, y + dt * k1 / 3)
            k3 = fn(t0 + dt * 2 / 3, y + dt * (k2 - k1 / 3))
            k4 = fn(t1, y + dt * (k1 - k2 + k3))
            y = y + dt * (k1 + 3 * (k2 + k3) + k4) / 8
        else:
            raise ValueError(method)
        ys.append(y)
    return torch.stack(ys)


# ---------------------------------------------------------------- E2TTS (e2_tts.py:1115-1595)

class E2TTS(Module):
    def __init__(self, transformer=None, duration_predictor=None,
                 odeint_kwargs=dict(atol=1e-5, rtol=1e-5, method='midpoint'), cond_drop_prob=0.25,
                 num_channels=None, mel_spec_module=None, num_freq_tokens=1, char_embed_kwargs=dict(),
                 mel_spec_kwargs=dict(), frac_lengths_mask=(0.7, 1.), concat_cond=False, interpolated_text=False,
                 text_num_embeds=None, tokenizer='char_utf8', use_vocos=False, pretrained_vocos_path=None,
                 sampling_rate=None, velocity_consistency_weight=0.):
        super().__init__()
        assert num_freq_tokens > 0 and not use_vocos
        self.num_freq_tokens, self.has_freq_axis = num_freq_tokens, num_freq_tokens > 1       # e2_tts.py:1150-1164
        if isinstance(transformer, dict):
            transformer = dict(transformer)
            transformer.setdefault('has_freq_axis', self.has_freq_axis)
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
        self.mel_spec = default(mel_spec_module, MelSpec(**mel_spec_kwargs))
        num_channels = default(num_channels, self.mel_spec.n_mel_channels)
        self.num_channels = num_channels
        self.sampling_rate = default(sampling_rate, getattr(self.mel_spec, 'sampling_rate', None))
        self.concat_cond = concat_cond                    # e2_tts.py:1196-1204
        if concat_cond:
            self.proj_in = nn.Linear(num_channels * 2, dim * num_freq_tokens)
        else:
            self.proj_in = nn.Linear(num_channels, dim * num_freq_tokens)
            self.cond_proj_in = nn.Linear(num_channels, dim * num_freq_tokens)
        self.maybe_split_freq = SplitFreq(num_freq_tokens) if self.has_freq_axis else nn.Identity()       # e2_tts.py:1208-1210
        self.to_pred = nn.Linear(dim, num_channels)
        if callable(tokenizer):
            assert exists(text_num_embeds)
            self.tokenizer = tokenizer
        elif tokenizer == 'char_utf8':
            text_num_embeds = 256
            self.tokenizer = list_str_to_tensor
        else:
            raise ValueError(f'unknown tokenizer string {tokenizer}')
        self.cond_drop_prob = cond_drop_prob
        embed_klass = InterpolatedCharacterEmbed if interpolated_text else CharacterEmbed      # e2_tts.py:1236-1238
        self.embed_text = embed_klass(dim_text, num_embeds=text_num_embeds, **char_embed_kwargs)
        self.register_buffer('zero', torch.tensor(0.), persistent=False)
        self.velocity_consistency_weight = velocity_consistency_weight

    @property
    def device(self):
        return next(self.parameters()).device

    def transformer_with_pred_head(self, x, cond, times, mask=None, text=None, drop_text_cond=None,
                                   return_drop_text_cond=False):
        seq_len = x.shape[-2]
        drop_text_cond = default(drop_text_cond, self.training and _pyrandom.random() < self.cond_drop_prob)
        if self.concat_cond:                              # e2_tts.py:1263-1276
            x = self.maybe_split_freq(self.proj_in(torch.cat((cond, x), dim=-1)))
        else:
            x = self.maybe_split_freq(self.proj_in(x)) + self.maybe_split_freq(self.cond_proj_in(cond))
        text_embed = None
        if exists(text) and not drop_text_cond:
            text_embed = self.embed_text(text, seq_len, mask=mask)
        embed = self.transformer(x, times=times, mask=mask, text_embed=text_embed)
        if self.has_freq_axis:
            embed = embed.mean(dim=1)
        pred = self.to_pred(embed)
        if not return_drop_text_cond:
            return pred
        return pred, drop_text_cond

    def cfg_transformer_with_pred_head(self, *args, cfg_strength=1., cfg_null_model=None,
                                       remove_parallel_component=True, keep_parallel_frac=0., **kwargs):
        pred = self.transformer_with_pred_head(*args, drop_text_cond=False, **kwargs)
        if cfg_strength < 1e-5:
            return pred
        null_drop = not exists(cfg_null_model)
        cfg_null_model = default(cfg_null_model, self)
        null_pred = cfg_null_model.transformer_with_pred_head(*args, drop_text_cond=null_drop, **kwargs)
        cfg_update = pred - null_pred
        if remove_parallel_component:
            parallel, orthogonal = project(cfg_update, pred)
            cfg_update = orthogonal + parallel * keep_parallel_frac
        return pred + cfg_update * cfg_strength

    @torch.no_grad()
    def sample(self, cond, *, text=None, lens=None, duration=None, steps=32, cfg_strength=1.,
               cfg_null_model=None, max_duration=4096, vocoder=None, return_raw_output=None,
               save_to_filename=None, _y0=None):
        self.eval()
        if cond.ndim == 2:
            cond = self.mel_spec(cond).transpose(1, 2)
            assert cond.shape[-1] == self.num_channels
        batch, cond_seq_len, device = cond.shape[0], cond.shape[1], cond.device
        if not exists(lens):
            lens = torch.full((batch,), cond_seq_len, device=device, dtype=torch.long)
        if isinstance(text, list):
            text = self.tokenizer(text).to(device)
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

        def fn(t, x):
            step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))
            return self.cfg_transformer_with_pred_head(x, step_cond, times=t, text=text, mask=mask,
                                                       cfg_strength=cfg_strength, cfg_null_model=cfg_null_model)

        y0 = _y0 if exists(_y0) else torch.randn_like(cond)
        t = torch.linspace(0, 1, steps, device=device)
        trajectory = odeint_fixed(fn, y0, t, self.odeint_kwargs.get('method', 'midpoint'))
        out = torch.where(cond_mask, cond, trajectory[-1])
        return out

    def forward(self, inp, *, text=None, times=None, lens=None, velocity_consistency_model=None,
                velocity_consistency_delta=1e-5, _noise=None):
        """_noise (test hook): dict with any of x0, times, frac_lengths, span_rand, drop_text_cond."""
        _noise = default(_noise, {})
        need_velocity_loss = exists(velocity_consistency_model) and self.velocity_consistency_weight > 0.      # e2_tts.py:1478
        if inp.ndim == 2:
            inp = self.mel_spec(inp).transpose(1, 2)
            assert inp.shape[-1] == self.num_channels
        batch, seq_len, dtype, device = inp.shape[0], inp.shape[1], inp.dtype, self.device
        if isinstance(text, list):
            text = self.tokenizer(text).to(device)
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
        w = (1. - t) * x0 + t * x1
        flow = x1 - x0
        cond = torch.where(rand_span_mask[..., None], torch.zeros_like(x1), x1)
        pred, did_drop = self.transformer_with_pred_head(w, cond, times=times, text=text, mask=mask,
                                                         drop_text_cond=_noise.get('drop_text_cond'),
                                                         return_drop_text_cond=True)
        velocity_loss = self.zero
        if need_velocity_loss:                      # e2_tts.py:1558-1576: EMA teacher at t + delta, same text-drop decision
            t_d = t + velocity_consistency_delta
            w_d = (1. - t_d) * x0 + t_d * x1
            with torch.no_grad():
                ema_pred = velocity_consistency_model.transformer_with_pred_head(
                    w_d, cond, times=times + velocity_consistency_delta, text=text, mask=mask, drop_text_cond=did_drop)
            velocity_loss = F.mse_loss(pred, ema_pred, reduction='none')
            velocity_loss = velocity_loss[rand_span_mask].mean()
        loss = F.mse_loss(pred, flow, reduction='none')
        loss = loss[rand_span_mask].mean()
        total_loss = loss + velocity_loss * self.velocity_consistency_weight
        return E2TTSReturn(total_loss, cond, pred, x0 + pred, LossBreakdown(loss, velocity_loss))
