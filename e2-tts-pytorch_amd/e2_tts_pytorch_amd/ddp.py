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

Two ways in:
  * `DataParallel(model)`: the wrapper used by bench.py -- call it like the model, then `loss.backward()`;
  * `enable_overlap_under_ddp(model)` + the stock `torch.nn.parallel.DistributedDataParallel` (what
    `accelerator.prepare(model)` builds, trainer.py:155-162,190): the backbone's parameters are taken out of the stock
    reducer (`_ddp_params_and_buffers_to_ignore`) and exchanged by the slab hook with overlap; everything else stays with
    stock DDP.  Without the shim stock DDP still works (the gradients travel through autograd), but all of its buckets
    become ready at once when the backbone's single autograd node finishes, i.e. nothing overlaps.

`grad_dtype=torch.bfloat16` halves the bytes on the links (2.9 GB -> 1.45 GB per dim-1024 / depth-24 step): a slab is
pre-divided by the world size in fp32, rounded to bf16, summed by RCCL in bf16 and added back into the fp32 buffer;
`wire_fp32_sum=True` keeps the bf16 wire but sums in fp32 (all-to-all of shards + local fp32 sum + all-gather: the same
bytes, one rounding instead of world - 1).
`bucket_layers=k` merges k consecutive layer slabs into one collective.

Hardware queues: the step runs on the caller's stream, the backbone's two launch lanes, this module's side stream and
RCCL's own streams.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); with a communicator
alive the fifth stream shares a queue with a launch lane and serialises against it (MI355X, cfg3: 93.3 -> 104.3 ms per
step from creating the communicator alone; 93.3 again with GPU_MAX_HW_QUEUES=8, profiles/r03_hw_queues.jsonl).  Set
GPU_MAX_HW_QUEUES=8 in the environment before the process makes its first device call (bench.py does).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from .backbone import Transformer


class _GradSync:
    def __init__(self, group=None, grad_dtype=torch.float32, bucket_layers=1, defer=False, wire_fp32_sum=False):
        assert grad_dtype in (torch.float32, torch.bfloat16)
        # bf16 wire with fp32 accumulation: RCCL's all-reduce sums in the wire dtype, i.e. an 8-rank bf16 sum rounds 7 times.  With
        # wire_fp32_sum the slab travels as bf16 shards (all-to-all = the reduce-scatter half), every rank sums ITS shard in fp32,
        # rounds once and the shards are all-gathered: the same bytes on the links as the bf16 all-reduce, one rounding
        self.wire_fp32_sum = bool(wire_fp32_sum) and grad_dtype == torch.bfloat16
        self.defer = bool(defer)      # one collective over everything when the backward pass has finished (no overlap, no CU sharing)
        self.group = group
        self.world = dist.get_world_size(group)
        self.grad_dtype = grad_dtype
        self.bucket_layers = max(1, int(bucket_layers))
        self.side = None
        self.lanes = []               # side streams of the backbone's launch lanes (set by the backbone): a slab is final on all of them
        self.calls = 0
        self.bytes = 0
        self._pending = None          # (start, end, layers) of the slabs merged so far
        self._wire = None             # bf16 wire buffer, grown to the largest slab seen (slabs are serialised on the side stream)
        self._recv = self._mine = None    # wire_fp32_sum: receive buffer (world copies of this rank's shard) and the summed shard
        be = dist.get_backend(group)
        self._avg = dist.ReduceOp.AVG if be == 'nccl' else None      # RCCL averages on the links; gloo (CPU tests) only sums

    def _reduce(self, gflat, start, end):
        slab = gflat[start:end]
        self.calls += 1
        self.bytes += slab.numel() * (2 if self.grad_dtype == torch.bfloat16 else 4)

        def run():
            if self.grad_dtype == torch.bfloat16:
                from . import ops
                if slab.is_cuda or ops.host_ok():
                    # one HIP pass each way through a preallocated wire buffer (no tensor-library temporaries on the side
                    # stream): pre-divided in fp32, rounded to bf16, summed by RCCL in bf16, widened back into the slab
                    # (wire_fp32_sum: the buffer is a whole number of 8-element-aligned shards, so that the all-to-all needs no padded copy)
                    per = self._shard_len(slab.numel())
                    need = per * self.world if self.wire_fp32_sum else slab.numel()
                    if self._wire is None or self._wire.numel() < need or self._wire.device != slab.device:
                        self._wire = torch.empty(need, dtype=torch.bfloat16, device=slab.device)
                        self._recv = self._mine = None
                    buf = self._wire[:slab.numel()]
                    ops.grad_pack_bf16(slab, buf, 1.0 / self.world)
                    if self.wire_fp32_sum and self.world > 1:
                        buf = self._sum_fp32(buf, kernels=True)
                    else:
                        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                    ops.grad_unpack_bf16(buf, slab)
                else:
                    buf = (slab * (1.0 / self.world)).to(torch.bfloat16)
                    if self.wire_fp32_sum and self.world > 1:
                        buf = self._sum_fp32(buf)
                    else:
                        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                    slab.copy_(buf)
            elif self._avg is not None:
                dist.all_reduce(slab, op=self._avg, group=self.group)          # in place, no scaling pass
            else:
                slab.mul_(1.0 / self.world)
                dist.all_reduce(slab, op=dist.ReduceOp.SUM, group=self.group)

        if gflat.is_cuda:
            if self.side is None:
                self.side = torch.cuda.Stream(device=gflat.device)
            self.side.wait_stream(torch.cuda.current_stream(gflat.device))     # the slab is complete on the compute stream
            for ls in self.lanes:                                               # ... and on the text / weight-gradient lanes
                self.side.wait_stream(ls)
            with torch.cuda.stream(self.side):
                run()
            slab.record_stream(self.side)
        else:
            run()

    def _shard_len(self, n):
        """elements per rank of a slab of n elements: rounded up to 8 (the kernels move 16 bytes per lane)"""
        per = (n + self.world - 1) // self.world
        return (per + 7) // 8 * 8

    def _sum_fp32(self, buf, kernels=False):
        """bf16 wire, fp32 sum: all-to-all of the world's shards, local fp32 sum of this rank's shard, one rounding, all-gather.
        -> a bf16 tensor of buf's length holding the sum (identical on every rank).  kernels: `buf` is the head of the preallocated
        wire buffer (per * world elements): no temporaries -- receive and shard buffers are kept, sized to the largest slab (round 6:
        five slab-sized allocations per slab on the exchange stream before), and the sum is one e2k_shard_sum_bf16 pass"""
        w, n = self.world, buf.numel()
        if kernels:
            from . import ops
            per = self._shard_len(n)
            send = self._wire[:per * w]
            if per * w > n:
                send[n:].zero_()                   # (the pad of the last shard: a handful of elements)
            if self._recv is None or self._recv.numel() < per * w:
                self._recv = torch.empty(self._wire.numel(), dtype=torch.bfloat16, device=buf.device)
                self._mine = torch.empty((self._wire.numel() // w + 7) // 8 * 8, dtype=torch.bfloat16, device=buf.device)
            recv, mine = self._recv[:per * w], self._mine[:per]
            dist.all_to_all_single(recv, send, group=self.group)      # recv[r * per : (r + 1) * per] = rank r's copy of MY shard
            ops.shard_sum_bf16(recv, mine, w)
            dist.all_gather_into_tensor(send, mine, group=self.group)  # the wire buffer takes the result: send is consumed by then
            return send[:n]
        per = (n + w - 1) // w
        send = buf if per * w == n else torch.cat([buf, buf.new_zeros(per * w - n)])
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)          # recv[r * per : (r + 1) * per] = rank r's copy of MY shard
        mine = recv.view(w, per).float().sum(0).to(torch.bfloat16)
        out = torch.empty_like(send)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        return out[:n]

    # Which parameters received a gradient is a GLOBAL fact under data parallelism.  The classifier-free-guidance coin that
    # drops the text stream (e2_tts.py:1261-1262) is flipped per rank; the reference's DistributedDataParallel
    # (find_unused_parameters=True, trainer.py:155-162) all-reduces its used-parameter map, so a parameter is updated on
    # every rank as soon as ANY rank used it and skipped on all of them otherwise.  The slab all-reduce gives every rank the
    # averaged text-stream gradients whatever its own coin said; the optimizer's skip decision (optim.FusedAdopt, and the
    # `None` gradients handed to autograd) must follow the same global fact or the replicas drift apart.  One MAX
    # all-reduce of a single word per forward pass, launched on the exchange stream when the forward starts and read when
    # the backward starts (long finished by then: no stall beyond the ranks' host skew).
    def begin_text_live(self, live, device):
        if self.world == 1:
            return bool(live)
        t = torch.full((1,), 1 if live else 0, dtype=torch.int32, device=device)     # (a fill kernel: torch.tensor(..., device=cuda) is a pageable host copy, which synchronises the host with the stream)
        if not t.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            return bool(t.item())
        if self.side is None:
            self.side = torch.cuda.Stream(device=device)
        self.side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(self.side):
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.side)
        t.record_stream(self.side)
        return (host, ev)

    @staticmethod
    def end_text_live(handle):
        if isinstance(handle, bool):
            return handle
        host, ev = handle
        ev.synchronize()
        return bool(host.item())

    def __call__(self, gflat, start, end):
        if start is None:                                   # flush, then wait for every slab launched so far
            if self._pending is not None:
                s, e, _ = self._pending
                self._pending = None
                self._reduce(gflat, s, e)
            if self.side is not None:
                torch.cuda.current_stream(gflat.device).wait_stream(self.side)
            return
        if end <= start:
            return
        if self.defer:                                      # just remember the covered range; reduced at the flush
            p0 = self._pending
            self._pending = (start, end, 1) if p0 is None else (min(p0[0], start), max(p0[1], end), p0[2] + 1)
            return
        # the backward finishes layers from the last to the first: consecutive slabs are adjacent in the flat buffer
        if self._pending is not None and self._pending[0] == end:
            s, e, k = start, self._pending[1], self._pending[2] + 1
        else:
            if self._pending is not None:
                self._reduce(gflat, self._pending[0], self._pending[1])
            s, e, k = start, end, 1
        if k >= self.bucket_layers:
            self._pending = None
            self._reduce(gflat, s, e)
        else:
            self._pending = (s, e, k)


def _null_ctx():
    import contextlib
    return contextlib.nullcontext()


def _warn_hw_queues():
    import os
    import warnings
    if torch.cuda.is_available() and int(os.environ.get('GPU_MAX_HW_QUEUES', '4')) < 6:
        warnings.warn('GPU_MAX_HW_QUEUES is below 6: the launch lanes, the gradient-exchange stream and RCCL will share hardware '
                      'queues and serialise (measured +11 ms per cfg3 step); export GPU_MAX_HW_QUEUES=8 before the first device call')


def _flat_params(bb):
    return [p for p, _ in bb._layout.slots]


@torch.no_grad()
def _broadcast(tensors, src, group):
    for t in tensors:
        dist.broadcast(t.data, src=src, group=group)


class DataParallel(nn.Module):
    """wraps an E2TTS / DurationPredictor / Transformer; call it like the wrapped module, then loss.backward()."""

    def __init__(self, module: nn.Module, process_group=None, broadcast_from: int | None = 0,
                 grad_dtype: torch.dtype = torch.float32, bucket_layers: int = 1, defer: bool = False, wire_fp32_sum: bool = False):
        super().__init__()
        assert dist.is_initialized(), 'torch.distributed must be initialised (backend "nccl" is RCCL on ROCm)'
        _warn_hw_queues()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._backbones = [m for m in module.modules() if isinstance(m, Transformer)]
        inside = {id(p) for bb in self._backbones for p in _flat_params(bb)}
        self._outside = [p for p in module.parameters() if id(p) not in inside]
        self._sync = _GradSync(process_group, grad_dtype, bucket_layers, defer, wire_fp32_sum)
        self._outside_queued = False
        for bb in self._backbones:
            bb._grad_sync = self._hook
        if broadcast_from is not None:
            _broadcast(list(module.parameters()) + list(module.buffers()), broadcast_from, process_group)
            for bb in self._backbones:          # the broadcast wrote behind the version counters: refresh the bf16 shadows
                bb._shadow_key = None

    @property
    def lanes(self):
        """side streams of the backbone's launch lanes: a slab is final only when they have caught up (set by the backbone)"""
        return self._sync.lanes

    @lanes.setter
    def lanes(self, streams):
        self._sync.lanes = list(streams)

    def begin_text_live(self, live, device):
        return self._sync.begin_text_live(live, device)

    def end_text_live(self, handle):
        return self._sync.end_text_live(handle)

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
        dev = ps[0].device
        side = None
        if dev.type == 'cuda':
            # on the exchange stream, like the slabs: the collective queues behind the last slab instead of in front of whatever the
            # compute stream does next; the compute stream waits for it once, at the end
            if self._sync.side is None:
                self._sync.side = torch.cuda.Stream(device=dev)
            side = self._sync.side
            side.wait_stream(torch.cuda.current_stream(dev))
        with (torch.cuda.stream(side) if side is not None else _null_ctx()):
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in ps])
            flat.mul_(1.0 / self.world)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            off = 0
            fresh = []               # gradients ALLOCATED in this block, i.e. by the exchange stream's pool
            for p in ps:
                n = p.numel()
                g = flat[off:off + n].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                    fresh.append(p.grad)
                else:
                    p.grad.copy_(g)
                off += n
        if side is not None:
            main = torch.cuda.current_stream(dev)
            fresh_ids = {id(g) for g in fresh}
            for p in ps:
                if id(p.grad) in fresh_ids:
                    # born on the exchange stream, read by the optimizer on the compute stream: the CONSUMER is what the allocator has to
                    # be told about, or zero_grad(set_to_none) hands the block back to the side pool while compute-stream readers are pending
                    p.grad.record_stream(main)
                else:
                    p.grad.record_stream(side)      # born on the compute stream, written here
            main.wait_stream(side)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)


def enable_overlap_under_ddp(module: nn.Module, process_group=None, broadcast_from: int | None = 0,
                             grad_dtype: torch.dtype = torch.float32, bucket_layers: int = 1, wire_fp32_sum: bool = False):
    """Call BEFORE wrapping `module` in the stock DistributedDataParallel (`accelerator.prepare(model)` in the reference
    trainer): every backbone's parameters are marked to be ignored by the stock reducer and exchanged by the per-layer
    slab hook instead (overlapped with the backbone's backward on a side stream); the remaining parameters keep going
    through stock DDP's buckets.  Returns the hook object (`.calls`, `.bytes` for inspection)."""
    assert dist.is_initialized()
    _warn_hw_queues()
    backbones = [(name, m) for name, m in module.named_modules() if isinstance(m, Transformer)]
    sync = _GradSync(process_group, grad_dtype, bucket_layers, wire_fp32_sum=wire_fp32_sum)
    ignore = list(getattr(module, '_ddp_params_and_buffers_to_ignore', []))
    names = {id(p): n for n, p in module.named_parameters()}
    for _, bb in backbones:
        bb._grad_sync = sync
        for p in _flat_params(bb):
            ignore.append(names[id(p)])
    module._ddp_params_and_buffers_to_ignore = ignore
    if broadcast_from is not None:               # stock DDP only broadcasts what it manages
        _broadcast([p for _, bb in backbones for p in _flat_params(bb)], broadcast_from, process_group)
        for _, bb in backbones:
            bb._shadow_key = None
    return sync
