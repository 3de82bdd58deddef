"""TEST INFRASTRUCTURE: constructor-independent weights for the golden vectors.

The reference model and the oracle (and the HIP host mirror) draw their initial weights in different orders, so a golden
file would have to carry every weight (tens of MB).  Instead every parameter is overwritten, in sorted-name order, from one
seeded generator: any model with the same state_dict keys and shapes gets the same values, and tests/golden/ only
stores seeds, inputs and the reference's outputs.  Buffers (rotary frequencies, mel filter bank, Hann window) are
deterministic functions of the configuration and are left alone, except the random Fourier frequencies of the time embedding.
"""
import torch


@torch.no_grad()
def fill_params(model, seed):
    g = torch.Generator().manual_seed(seed)
    for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
        if 'dynamic_alpha_fn' in name or 'dynamic_beta_fn' in name:
            # hyper-connection mixing projections (dim, s + 1) / (dim,): unit-variance pre-activations, so that the tanh is
            # not saturated and gradients stay well-conditioned (saturated, bf16 rounding of the weights alone moves the
            # input gradient of the fp32 model by 17 %)
            v = torch.randn(p.shape, generator=g) * (p.shape[0] ** -0.5)
        elif p.ndim >= 2:
            v = torch.randn(p.shape, generator=g) * (0.5 / p.shape[-1] ** 0.5)
        else:
            v = torch.randn(p.shape, generator=g) * 0.3
            if name.endswith('.g') or name.endswith('static_beta'):      # norm gains / branch weights: around one
                v = v + 1.
        p.copy_(v.to(p.dtype))
    for name, b in sorted(model.named_buffers(), key=lambda kv: kv[0]):
        if name.endswith('.weights'):          # RandomFourierEmbed (e2_tts.py:355-364): the one buffer that is drawn at random
            b.copy_(torch.randn(b.shape, generator=g))
    return model
