"""Dataset side of the hot path ("GPU-side dataset MelSpec batching").

The reference computes the log-mel of every clip on the CPU inside `HFDataset.__getitem__` (trainer.py:101-131, one
torchaudio transform per clip) and `collate_fn` (trainer.py:61-82) zero-pads the spectrograms to the longest one.
Here the data loader only pads raw samples; the whole batch is transformed on the GPU in one launch of the ragged MelSpec
kernel, which treats every row as if it were alone (reflection about its own end, its own frame count, zeros after
it).  `mel_batch` returns what the reference's `collate_fn` returns, so the trainer's step

    batch = collate_wave_fn(items)                          # in the DataLoader workers: no FFT on the CPU
    batch = mel_batch(batch, mel_spec, device)              # one kernel launch
    loss = model(batch['mel'], text=batch['text'], lens=batch['mel_lengths'])

is unchanged from there on (trainer.py:258-263 permutes `mel` to (b, n, d) itself).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


class Resample(nn.Module):
    """torchaudio.transforms.Resample(orig_freq, new_freq) (defaults: Hann-windowed sinc, lowpass_filter_width 6, rolloff 0.99) as
    `HFDataset.__getitem__` applies it to a clip whose sample rate is not the model's (trainer.py:116-118) -- here for a whole
    zero-padded batch on the device (e2k_resample_sinc): forward(wave (b, n) or (n,), lens=None) -> wave at the new rate,
    (b, ceil(n new / orig)); with `lens` (valid samples per row) also the rows' new lengths ceil(len new / orig), and every row is
    converted as if it were alone.  The polyphase filter bank (new / gcd filters of 2 width + orig / gcd taps) is built once, in
    fp64, on the host."""

    def __init__(self, orig_freq=16000, new_freq=16000, lowpass_filter_width=6, rolloff=0.99):
        super().__init__()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        g = math.gcd(self.orig_freq, self.new_freq)
        self.orig, self.new = self.orig_freq // g, self.new_freq // g
        self.width = 0
        if self.orig != self.new:
            base = min(self.orig, self.new) * rolloff
            self.width = math.ceil(lowpass_filter_width * self.orig / base)
            idx = torch.arange(-self.width, self.width + self.orig, dtype=torch.float64)[None] / self.orig
            t = torch.arange(0, -self.new, -1, dtype=torch.float64)[:, None] / self.new + idx
            t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
            window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
            t = t * math.pi
            k = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / self.orig)
            self.register_buffer('kernel', k.float().contiguous(), persistent=False)          # (new, 2 width + orig)

    def new_length(self, n):
        return (n * self.new + self.orig - 1) // self.orig

    def forward(self, wave, lens=None):
        if self.orig == self.new:
            return wave if lens is None else (wave, lens)
        one = wave.ndim == 1
        x = wave.reshape(-1, wave.shape[-1]).float()
        if x.stride(-1) != 1:
            x = x.contiguous()
        if self.kernel.device != x.device:
            self.to(x.device)
        l32 = None if lens is None else lens.to(device=x.device, dtype=torch.int32).contiguous()
        out = ops.resample_sinc(x, self.kernel, self.orig, self.new, self.width, l32)
        out = out[0] if one else out.reshape(wave.shape[:-1] + out.shape[-1:])
        if lens is None:
            return out
        return out, (lens * self.new + self.orig - 1) // self.orig


def collate_wave_fn(batch):
    """items: dicts with 'wave' (1-D float tensor or array) and 'text'; optionally 'sampling_rate' (as the rows of the reference's
    datasets carry it, trainer.py:107): passed through as a LongTensor so that mel_batch can convert clips that are not at the model's
    rate (on the device, for the batch) instead of the data loader workers doing it clip by clip"""
    waves = [torch.as_tensor(item['wave'], dtype=torch.float32).reshape(-1) for item in batch]
    wave_lengths = torch.LongTensor([w.shape[0] for w in waves])
    n = int(wave_lengths.amax())
    wave = torch.stack([F.pad(w, (0, n - w.shape[0])) for w in waves])
    text = [item['text'] for item in batch]
    out = dict(wave=wave, wave_lengths=wave_lengths, text=text, text_lengths=torch.LongTensor([len(t) for t in text]))
    if all('sampling_rate' in item for item in batch):
        out['sampling_rates'] = torch.LongTensor([int(item['sampling_rate']) for item in batch])
    return out


_resamplers = {}


def mel_batch(batch, mel_spec, device=None):
    """-> dict(mel (b, n_mels, frames), mel_lengths, text, text_lengths): the reference collate_fn's output.  Rows whose
    'sampling_rates' entry differs from mel_spec.sampling_rate are resampled first (trainer.py:116-118), all rows of one source rate
    in one launch"""
    wave, wl = batch['wave'], batch['wave_lengths']
    if device is not None:
        wave, wl = wave.to(device, non_blocking=True), wl.to(device, non_blocking=True)
    rates = batch.get('sampling_rates')
    target = int(mel_spec.sampling_rate)
    if rates is not None and any(int(r) != target for r in rates.tolist()):
        parts, lens = [None] * wave.shape[0], wl.clone()
        for r in sorted(set(rates.tolist())):
            rows = [i for i, v in enumerate(rates.tolist()) if v == r]
            idx = torch.tensor(rows, device=wave.device)
            if r == target:
                w, l = wave[idx], wl[idx]
            else:
                rs = _resamplers.get((r, target))
                if rs is None:
                    rs = _resamplers[(r, target)] = Resample(r, target)
                w, l = rs(wave[idx], lens=wl[idx])
            for k, i in enumerate(rows):
                parts[i] = w[k]
            lens[idx] = l
        n = max(int(p.shape[0]) for p in parts)
        wave = torch.stack([F.pad(p, (0, n - p.shape[0])) for p in parts])
        wl = lens
    mel = mel_spec(wave, lens=wl)
    return dict(mel=mel, mel_lengths=1 + wl // mel_spec.hop_length, text=batch['text'], text_lengths=batch['text_lengths'])
