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

import torch
import torch.nn.functional as F


def collate_wave_fn(batch):
    """items: dicts with 'wave' (1-D float tensor or array, already at the target sample rate) and 'text'"""
    waves = [torch.as_tensor(item['wave'], dtype=torch.float32).reshape(-1) for item in batch]
    wave_lengths = torch.LongTensor([w.shape[0] for w in waves])
    n = int(wave_lengths.amax())
    wave = torch.stack([F.pad(w, (0, n - w.shape[0])) for w in waves])
    text = [item['text'] for item in batch]
    return dict(wave=wave, wave_lengths=wave_lengths, text=text, text_lengths=torch.LongTensor([len(t) for t in text]))


def mel_batch(batch, mel_spec, device=None):
    """-> dict(mel (b, n_mels, frames), mel_lengths, text, text_lengths): the reference collate_fn's output"""
    wave, wl = batch['wave'], batch['wave_lengths']
    if device is not None:
        wave, wl = wave.to(device, non_blocking=True), wl.to(device, non_blocking=True)
    mel = mel_spec(wave, lens=wl)
    return dict(mel=mel, mel_lengths=1 + wl // mel_spec.hop_length, text=batch['text'], text_lengths=batch['text_lengths'])
