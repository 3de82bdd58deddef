"""TEST INFRASTRUCTURE: the fp32 oracle with its intermediate tensors rounded to bf16.

How far a gradient of the fp32 model moves when activations and activation gradients are rounded to bf16 at the places
where a bf16 implementation has to store them (after every Linear / norm / GLU / conv / gate / softmax and at the
hyper-connection boundaries) measures how well-conditioned that gradient is.  Parity tests use it to size the tolerance
of quantities that are ill-conditioned in the MODEL (not in the kernels): the HIP path may deviate from the fp32 oracle
by a small multiple of what this emulation already deviates by.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import e2tts_oracle as O


class RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class RoundFwd(torch.autograd.Function):
    """bf16 copy of a parameter as a matrix-unit operand: the forward (and, through autograd, the dgrad) product sees the rounded
    weight, its gradient stays fp32 -- what a bf16 shadow of an fp32 master weight does"""
    @staticmethod
    def forward(ctx, w):
        return w.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class bf16_intermediates:
    """context manager: oracle modules round their outputs (and the gradients flowing back through them) to bf16"""

    def __enter__(self):
        self.mods = [O.RMSNorm, O.AdaptiveRMSNorm, O._GLU, O.DepthwiseConv, O.AdaLNZero, O._HCNorm]
        self.saved = {m: m.forward for m in self.mods}
        self.hc0, self.lin0, self.sm0 = O.HyperConnections.forward, nn.Linear.forward, torch.Tensor.softmax
        hc0, lin0, sm0 = self.hc0, self.lin0, self.sm0

        def hc(self_, residuals):
            b, add = hc0(self_, RoundBoth.apply(residuals))
            return RoundBoth.apply(b), (lambda y: RoundBoth.apply(add(y)))

        O.HyperConnections.forward = hc
        # (round 6: the Linear's weight as a bf16 operand too -- every GEMM of a bf16 implementation reads a rounded copy of the fp32
        #  master weight; before, only activations were rounded and the emulation under-stated the noise of the gradients by ~1.7 x)
        nn.Linear.forward = lambda self_, x: RoundBoth.apply(F.linear(RoundBoth.apply(x), RoundFwd.apply(self_.weight), self_.bias))
        torch.Tensor.softmax = lambda self_, *a, **k: RoundBoth.apply(sm0(self_, *a, **k))
        for m in self.mods:
            m.forward = (lambda f: lambda self_, *a, **k: RoundBoth.apply(f(self_, *a, **k)))(self.saved[m])
        return self

    def __exit__(self, *exc):
        O.HyperConnections.forward, nn.Linear.forward, torch.Tensor.softmax = self.hc0, self.lin0, self.sm0
        for m in self.mods:
            m.forward = self.saved[m]
        return False
