"""Generates tests/golden/oracle_small.pt: seeded inputs + explicit noise + the CPU oracle's outputs.

These vectors pin the ORACLE only (drift detector + a fixture the GPU box can check without /root/reference); the
vectors that come from the reference's own source are tests/golden/reference_pinned.pt (oracle/pin_against_reference.py).
Run:  python tests/golden/make_golden.py
"""
import random
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
from oracle import e2tts_oracle as O  # noqa: E402
from test_backbone import randomize  # noqa: E402


def main():
    random.seed(11)
    torch.manual_seed(11)
    kw = dict(dim=256, depth=2, heads=4, dropout=0., max_seq_len=64)
    model = O.E2TTS(transformer=dict(**kw), cond_drop_prob=0.)
    randomize(model, seed=5)
    B, T = 2, 48
    mel = torch.randn(B, T, 100)
    lens = torch.tensor([T, 37])
    noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.tensor([0.8, 0.95]),
                 span_rand=torch.tensor([0.25, 0.5]), drop_text_cond=False)
    text = ['golden', 'vector test']
    out = model(mel, text=text, lens=lens, _noise=noise)
    out.loss.backward()
    wave = torch.randn(1, 256 * 9)
    # the weights are reproducible from the seeds (torch CPU RNG): only a checksum of them is stored
    wsum = sum(float(v.double().abs().sum()) for v in model.state_dict().values())
    fix = dict(kw=kw, seeds=(11, 5), weight_abs_sum=wsum, mel=mel, lens=lens, noise=noise,
               text=text, loss=out.loss.detach(), pred_flow=out.pred_flow.detach(), cond=out.cond,
               grad_to_pred=model.to_pred.weight.grad.clone(),
               grad_registers=model.transformer.registers.grad.clone(),
               wave=wave, logmel=O.MelSpec()(wave))
    torch.save(fix, Path(__file__).resolve().parent / 'oracle_small.pt')
    print('saved', float(out.loss))


if __name__ == '__main__':
    main()
