"""Data-parallel gradient exchange for the flat-gradient backbone: one process per GPU, RCCL over xGMI.

Replaces the implicit DistributedDataParallel reducer that `accelerate` sets up in the reference trainer
(/root/reference/e2_tts_pytorch/trainer.py:155-162,190-192,270).  The path shards by batch only: every rank
runs the full model on its own sequences and the only exchange is the mean of the gradients.  Because the
backbone's gradients already live in one flat fp32 buffer laid out layer by layer, the "buckets" are simply the
per-layer slabs (~30 M parameters = ~120 MB each at dim 1024): as soon as the hand-scheduled backward has
finished a layer, its slab is all-reduced on a side HIP stream while the backward of the earlier layers keeps
the compute stream busy -- few, large collectives, which is what point-to-point xGMI links want.
Parameters outside the backbone (100-channel projections, text embedding, time MLP; < 1 M) are reduced in one
small flat all-reduce when the autograd pass ends.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from .backbone import Transformer


class _GradSync:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.side = None
        self.calls = 0

    def __call__(self, gflat, start, end):
        if start is None:                                   # wait for every slab launched so far
            if self.side is not None:
                torch.cuda.current_stream(gflat.device).wait_stream(self.side)
            return
        if end <= start:
            return
        slab = gflat[start:end]
        self.calls += 1
        if gflat.is_cuda:
            if self.side is None:
                self.side = torch.cuda.Stream(device=gflat.device)
            self.side.wait_stream(torch.cuda.current_stream(gflat.device))     # the slab is complete on the compute stream
            with torch.cuda.stream(self.side):
                slab.mul_(1.0 / self.world)
                dist.all_reduce(slab, op=dist.ReduceOp.SUM, group=self.group)
            slab.record_stream(self.side)
        else:
            slab.mul_(1.0 / self.world)
            dist.all_reduce(slab, op=dist.ReduceOp.SUM, group=self.group)


class DataParallel(nn.Module):
    """wraps an E2TTS / DurationPredictor / Transformer; call it like the wrapped module, then loss.backward()."""

    def __init__(self, module: nn.Module, process_group=None, broadcast_from: int | None = 0):
        super().__init__()
        assert dist.is_initialized(), 'torch.distributed must be initialised (backend "nccl" is RCCL on ROCm)'
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._backbones = [m for m in module.modules() if isinstance(m, Transformer)]
        inside = {id(p) for bb in self._backbones for p in bb.parameters() if self._in_flat(bb, p)}
        self._outside = [p for p in module.parameters() if id(p) not in inside]
        self._sync = _GradSync(process_group)
        for bb in self._backbones:
            bb._grad_sync = self._hook
        self._outside_queued = False
        if broadcast_from is not None:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=broadcast_from, group=process_group)
            for bb in self._backbones:          # the broadcast wrote behind the version counters: refresh the bf16 shadows
                bb._shadow_key = None

    @staticmethod
    def _in_flat(bb, p):
        return any(p is q for q, _ in bb._layout.slots)

    def _hook(self, gflat, start, end):
        if start is None and self._outside and not self._outside_queued:
            # the rest of the autograd pass (input projections, embeddings) finishes after the backbone: reduce then
            # (once per backward pass, however many backbones the wrapped module holds)
            self._outside_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._reduce_outside)
        self._sync(gflat, start, end)

    def _reduce_outside(self):
        self._outside_queued = False
        # every rank contributes every parameter (zeros where it produced no gradient, e.g. the text embedding on a
        # step whose classifier-free-guidance coin flip dropped the text): the collective has the same size everywhere
        ps = [p for p in self._outside if p.requires_grad]
        if not ps:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in ps])
        flat.mul_(1.0 / self.world)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p in ps:
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)
