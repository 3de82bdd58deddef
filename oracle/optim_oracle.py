"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the optimizer-side steps of the reference trainer
(/root/reference/e2_tts_pytorch/trainer.py:272-279), PARITY UNPINNED: `adam_atan2_pytorch` and `ema_pytorch` are
un-vendored third-party packages that are not installed here; the arithmetic below restates SURVEY.md Appendix A.10 /
A.11 (themselves written from knowledge of those packages).  Plain torch fp32, one tensor list at a time.
"""
from __future__ import annotations

import torch


def clip_grad_norm_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ as accelerator.clip_grad_norm_ applies it (trainer.py:272-273)"""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


class Adopt:
    """adam_atan2_pytorch.adopt.Adopt(params, lr, betas=(0.9, 0.99), eps=1e-6, weight_decay=0, decoupled_wd=True)
    (trainer.py:183,275): step 0 only sets v = g^2; afterwards
    u = clamp(g / max(sqrt(v), eps), +-step^0.25), m.lerp_(u, 1 - beta1), p -= lr * m, v.lerp_(g^2, 1 - beta2); the weight decay
    p *= 1 - lr * (wd / init_lr if decoupled_wd else wd) is applied before all of that, on every step including the first.
    `steps` is per-parameter state and a parameter whose .grad is None is skipped (its count does not advance), as in
    the torch.optim-style loop of that package -- which matters here: the text stream has no gradient on the 25 % of the
    training steps whose classifier-free-guidance coin drops the text (e2_tts.py:1261-1262)."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.99), eps=1e-6, weight_decay=0., decoupled_wd=True):
        self.params = list(params)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.init_lr, self.decoupled_wd = lr, decoupled_wd      # (`lr` may be changed by a scheduler; the decoupling divides by the INITIAL one)
        self.steps = [0] * len(self.params)
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self):
        b1, b2 = self.betas
        for i, (p, m, v) in enumerate(zip(self.params, self.m, self.v)):
            g = p.grad
            if g is None:
                continue
            # weight decay first -- before the state exists, i.e. also on the step that only sets v -- and, decoupled, divided by the
            # initial learning rate (`wd /= init_lr; if wd > 0: p.mul_(1 - lr * wd)` in that package's step())
            wd = self.wd / self.init_lr if self.decoupled_wd else self.wd
            if wd > 0:
                p.mul_(1. - self.lr * wd)
            t = self.steps[i]
            self.steps[i] += 1
            if t == 0:
                v.copy_(g * g)
                continue
            u = (g / torch.clamp(v.sqrt(), min=self.eps)).clamp(-t ** 0.25, t ** 0.25)
            m.lerp_(u, 1. - b1)
            p.add_(m, alpha=-self.lr)
            v.lerp_(g * g, 1. - b2)

    @property
    def step_count(self):
        return max(self.steps) if self.steps else 0


def ema_decay(step, beta=0.9999, update_after_step=100, inv_gamma=1., power=2. / 3., min_value=0.):
    """ema_pytorch.EMA.get_current_decay"""
    epoch = max(step - update_after_step - 1, 0)
    if epoch <= 0:
        return 0.
    value = 1. - (1. + epoch / inv_gamma) ** -power
    return min(max(value, min_value), beta)
