"""Data-parallel path: 2 ranks over gloo on CPU (kernels = host logic-checker build).  The per-layer slab all-reduce
inside the hand-scheduled backward must give every rank the mean of the per-rank gradients, i.e. the gradient of the
mean loss over the global batch, for backbone and non-backbone parameters alike (also when one rank drops the text)."""
import os
import random
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def install_lib(path, host_pointers):
    from emu.install import install
    install(path, host_pointers)


ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, emu_lib, q):
    sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT), str(ROOT / 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), E2K_EMU_THREADS='2')
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from e2_tts_pytorch_amd import E2TTS, _lib
    from e2_tts_pytorch_amd.ddp import DataParallel
    from test_backbone import randomize
    install_lib(emu_lib, host_pointers=True)
    random.seed(7 + rank)                 # different init per rank: the wrapper must broadcast rank 0's weights
    torch.manual_seed(7 + rank)
    model = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0.), use_vocos=False, cond_drop_prob=0.)
    randomize(model, seed=rank)
    net = DataParallel(model)
    torch.manual_seed(100)
    B, T = 1, 24
    mels = torch.randn(world, B, T, 100)
    noises = [dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.tensor([0.8]),
                   span_rand=torch.tensor([0.4]), drop_text_cond=(r == 1)) for r in range(world)]
    out = net(mels[rank], text=['hello'], _noise=noises[rank])
    net._sync.lanes = None                # (must be set by the backbone through DataParallel.lanes before the first slab goes out)
    out.loss.backward()
    assert net._sync.lanes == []          # no streams on the host model, but the hand-over has happened
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    # persistent-gradient mode: the slabs are views of one long-lived buffer; three more steps (other inputs first, so
    # that the last one has something to overwrite) must end with the same averaged gradients
    tr = model.transformer
    tr.enable_persistent_grads()
    flat_ids = {id(q_) for q_, _ in tr._layout.slots}
    bad_p = []
    for which in ((rank + 1) % world, (rank + 1) % world, rank):      # first sighting (eager), recording, replay
        for p in model.parameters():
            if id(p) not in flat_ids:
                p.grad = None
        net(mels[which], text=['hello'], _noise=noises[rank]).loss.backward()
    for n, p in model.named_parameters():
        if n in grads:
            err = ((p.grad - grads[n]).norm() / grads[n].norm().clamp_min(1e-12)).item() if float(grads[n].norm()) > 0 else float(p.grad.norm())
            if err > 1e-4:
                bad_p.append((n + ' (persistent)', err))
    assert tr._pg is not None and all(q_.grad is v for (q_, _), v in zip(tr._layout.slots, tr._pg.views) if q_.requires_grad)
    # the last two steps were a plan RECORDING (second sighting of the signature) and a plan REPLAY, both with the launch
    # lanes on (rank 0; rank 1 drops the text, whose schedule is single-lane) and the slab hook firing between the recorded
    # backward segments: the combination the 8-GPU run uses
    plans = [v for v in tr._plans.values() if not isinstance(v, str)]
    assert len(plans) == 1 and plans[0].bwd and plans[0].segs and len(plans[0].lane_ss) == (2 if rank == 0 else 0) and tr._plan_tick >= 2, (len(plans), tr._plan_tick, bool(plans[0].bwd), plans[0].segs and len(plans[0].segs), plans[0].lane_ss)
    assert sum(1 for _f, _c, slab in plans[0].segs if slab is not None) >= 2          # hook points inside the replayed backward
    tr.enable_persistent_grads(False)
    # the stock torch DistributedDataParallel (what accelerator.prepare builds, trainer.py:155-162,190) with the overlap
    # shim: the backbone leaves the stock reducer, its slabs go through the hook -- here in bf16, two layers per collective
    from torch.nn.parallel import DistributedDataParallel as DDP
    from e2_tts_pytorch_amd.ddp import enable_overlap_under_ddp
    for bb in (tr,):
        bb._grad_sync = None
    model.zero_grad(set_to_none=True)
    hook = enable_overlap_under_ddp(model, grad_dtype=torch.bfloat16, bucket_layers=2, wire_fp32_sum=True)      # bf16 wire, fp32 sum (all-to-all + all-gather)
    stock = DDP(model, find_unused_parameters=True)
    stock(mels[rank], text=['hello'], _noise=noises[rank]).loss.backward()
    bad_s = []
    for n, p in model.named_parameters():
        if n in grads:
            g = grads[n]
            err = ((p.grad - g).norm() / g.norm().clamp_min(1e-12)).item() if float(g.norm()) > 0 else float(p.grad.norm())
            if err > (1e-2 if p.numel() > 1 else 1e-1):   # bf16 rounding of the summed slabs (scalars: the two ranks' values nearly cancel)
                bad_s.append((n + ' (stock DDP + shim, bf16 slabs)', err))
    assert hook.calls >= 2 and any(n.startswith('transformer.') for n in model._ddp_params_and_buffers_to_ignore)
    bad_p += bad_s
    # round 6: the wrapper with bucket_layers = 2 AND defer = True (ONE collective over everything the backward covered, issued at the
    # flush), bf16 wire: same averaged gradients, exactly one slab collective
    del stock
    tr._grad_sync = None
    model.zero_grad(set_to_none=True)
    net2 = DataParallel(model, broadcast_from=None, grad_dtype=torch.bfloat16, bucket_layers=2, defer=True)
    net2(mels[rank], text=['hello'], _noise=noises[rank]).loss.backward()
    assert net2._sync.calls == 1, net2._sync.calls
    for n, p in model.named_parameters():
        if n in grads:
            g = grads[n]
            err = ((p.grad - g).norm() / g.norm().clamp_min(1e-12)).item() if float(g.norm()) > 0 else float(p.grad.norm())
            if err > (1e-2 if p.numel() > 1 else 1e-1):
                bad_p.append((n + ' (bucket_layers 2 + defer, bf16 slab)', err))
    if rank == 0:
        # single-process reference with the broadcast weights: mean over both "ranks" of the per-sample gradients
        ref = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0.), use_vocos=False, cond_drop_prob=0.)
        ref.load_state_dict(model.state_dict())
        acc = {}
        for r in range(world):
            ref.zero_grad(set_to_none=True)
            o = ref(mels[r], text=['hello'], _noise=noises[r])
            o.loss.backward()
            for n, p in ref.named_parameters():
                if p.grad is not None:
                    acc[n] = acc.get(n, 0) + p.grad / world
        bad = []
        for n, g in acc.items():
            err = ((grads[n] - g).norm() / g.norm().clamp_min(1e-12)).item() if float(g.norm()) > 0 else float(grads[n].norm())
            if err > 2e-2:
                bad.append((n, err))
        q.put(('ok', bad + bad_p, net._sync.calls))
    else:
        q.put(('ok', bad_p, net._sync.calls))
    dist.barrier()
    dist.destroy_process_group()


def _sum_worker(rank, world, port, q, emu_lib=None):
    sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT), str(ROOT / 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), E2K_EMU_THREADS='1')
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from e2_tts_pytorch_amd.ddp import _GradSync
    sync = _GradSync(None, torch.bfloat16, 1, wire_fp32_sum=True)
    bad = []
    if emu_lib is not None:
        # the product path of the same exchange (round 6): pack kernel -> preallocated wire buffer of whole 8-aligned shards -> all-to-all ->
        # e2k_shard_sum_bf16 -> all-gather into the wire buffer -> unpack kernel; slabs of three lengths through ONE _GradSync (the buffers
        # are kept and sized to the largest), the largest first and last
        install_lib(emu_lib, host_pointers=True)
        for n in (4099, 1001, 24, 4099):
            gs = [torch.randn(n + 16, generator=torch.Generator().manual_seed(7 * n + r)) * 3 for r in range(world)]
            g = gs[rank].clone()
            sync._reduce(g, 8, 8 + n)
            want = sum((x[8:8 + n] * (1.0 / world)).to(torch.bfloat16).float() for x in gs).to(torch.bfloat16).float()
            if not torch.equal(g[8:8 + n], want):
                bad.append((n, 'kernel path: slab differs from the once-rounded fp32 sum of the pre-divided bf16 slabs'))
            if not (torch.equal(g[:8], gs[rank][:8]) and torch.equal(g[8 + n:], gs[rank][8 + n:])):
                bad.append((n, 'kernel path: wrote outside the slab'))
        assert sync._recv is not None and sync._wire.numel() >= 4099
    for n in (1001, 3 * 512, 5):                       # a length the world size does not divide, one it does, one shorter than a shard row
        bufs = [(torch.randn(n, generator=torch.Generator().manual_seed(10 * n + r)) * 3).to(torch.bfloat16) for r in range(world)]
        got = sync._sum_fp32(bufs[rank].clone())
        want = sum(b.float() for b in bufs).to(torch.bfloat16)          # ONE rounding of the fp32 sum
        chain = bufs[0].clone()
        for b in bufs[1:]:
            chain = (chain + b)                                         # what a bf16 all-reduce does: a rounding per addition
        if got.shape != (n,) or not torch.equal(got, want):
            bad.append((n, 'sum differs from the once-rounded fp32 sum'))
        if n > 100 and torch.equal(want, chain):
            bad.append((n, 'test is vacuous: the bf16 chain gives the same bits'))
    q.put(('ok', bad))
    dist.barrier()
    dist.destroy_process_group()


def test_wire_fp32_sum_rounds_once(emu_lib):
    """ddp._GradSync._sum_fp32 (wire_fp32_sum=True; replaces the bf16 all-reduce of the gradient slabs, trainer.py:155-162): bf16 on the
    wire, fp32 accumulation -- the result is bit for bit the fp32 sum of the ranks' bf16 slabs rounded ONCE, on every rank, also when the
    world size does not divide the slab.  3 ranks, gloo."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_sum_worker, args=(r, 3, port, q, str(emu_lib))) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[0] == 'ok' and not r[1] for r in res), res


def test_two_rank_gradient_mean(emu_lib):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(emu_lib), q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _queue
    res, waited = [], 0
    while len(res) < len(procs) and waited < 900:
        try:
            res.append(q.get(timeout=5))
        except _queue.Empty:
            waited += 5
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead:                                        # a rank died (assertion in the worker): do not wait for the other one
                for p in procs:
                    p.kill()
                raise AssertionError(f'a rank exited with {dead} (see its traceback above)')
    assert len(res) == len(procs), 'timed out'
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for status, bad, calls in res:
        assert status == 'ok' and not bad, bad[:10]
        assert calls >= 3          # one slab per layer + the global/conditioning slab


def _adopt_worker(rank, world, port, emu_lib, q):
    """the classifier-free-guidance coin is flipped per rank; whether the text stream's parameters are
    updated must be the same on every rank (the reference's DDP all-reduces its used-parameter map).  Three optimizer steps
    -- rank 1 drops the text, nobody does, everybody does -- through both gradient paths (persistent flat buffer = one fused
    launch; views handed to autograd = the per-run path) and under the stock DistributedDataParallel with the overlap shim: parameters, moments and step counts must be identical on the two
    ranks after every step, and on the all-dropped step the text stream must not move at all."""
    sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT), str(ROOT / 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), E2K_EMU_THREADS='2')
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from e2_tts_pytorch_amd import E2TTS
    from e2_tts_pytorch_amd.ddp import DataParallel
    from e2_tts_pytorch_amd.optim import FusedAdopt
    from test_backbone import randomize
    install_lib(emu_lib, host_pointers=True)
    bad = []
    # (wrapper, persistent flat gradient buffer): this package's DataParallel through both gradient paths, and the stock
    # DistributedDataParallel with the overlap shim (what accelerator.prepare builds around the reference trainer's model)
    # ... and the stock DDP WITHOUT the shim: the flag is then this rank's own until FusedAdopt.step makes it global
    for wrapper, persistent in (('dp', True), ('dp', False), ('stock', False), ('stock_no_shim', False)):
        random.seed(11)
        torch.manual_seed(11)
        model = E2TTS(transformer=dict(dim=256, depth=2, heads=4, dropout=0., num_registers=8), use_vocos=False, cond_drop_prob=0.)
        randomize(model, seed=3)
        if wrapper == 'dp':
            net = DataParallel(model)
        else:
            from torch.nn.parallel import DistributedDataParallel as DDP
            from e2_tts_pytorch_amd.ddp import enable_overlap_under_ddp
            if wrapper == 'stock':
                enable_overlap_under_ddp(model)
            net = DDP(model, find_unused_parameters=True)
        tr = model.transformer
        tr.enable_plans(False)
        tr.enable_persistent_grads(persistent)
        persistent = (wrapper, persistent)          # (label of the failure records below)
        opt = FusedAdopt(model, lr=1e-2, max_grad_norm=1.0)
        text_ids = tr._text_param_ids()
        names = {id(p): n for n, p in model.named_parameters()}
        torch.manual_seed(200 + rank)
        B, T = 1, 16
        for step, drops in enumerate(((False, True), (False, False), (True, True))):
            mel = torch.randn(B, T, 100)
            noise = dict(x0=torch.randn(B, T, 100), times=torch.rand(B), frac_lengths=torch.tensor([0.8]),
                         span_rand=torch.tensor([0.4]), drop_text_cond=drops[rank])
            before = {id(p): p.detach().clone() for p in model.parameters() if id(p) in text_ids}
            net(mel, text=['hello'], _noise=noise).loss.backward()
            opt.step()
            opt.zero_grad()
            sd = opt.state_dict()['state']
            flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()] +
                             [torch.cat([sd[i]['m'].reshape(-1), sd[i]['v'].reshape(-1)]) for i in sorted(sd)] +
                             [torch.tensor([float(s) for s in opt.steps])])
            both = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            if not torch.equal(both[0], both[1]):
                bad.append((persistent, step, 'ranks differ', float((both[0] - both[1]).abs().max())))
            moved = [names[i] for i, b in before.items() if not torch.equal(b, dict((id(p), p) for p in model.parameters())[i].detach())]
            if all(drops) and moved:
                bad.append((persistent, step, 'text stream moved although every rank dropped it', moved[:3]))
            if not all(drops) and step > 0 and len(moved) < len(before) // 2:          # (ADOPT's first step only initialises v)
                bad.append((persistent, step, 'text stream did not move although a rank used it', len(moved), len(before)))
        tcount = {opt.steps[opt._index[i]] for i in text_ids if i in opt._index}
        ocount = {s for i, s in enumerate(opt.steps) if id(opt.params[i]) not in text_ids and names[id(opt.params[i])].startswith('transformer.')}
        if tcount != {2} or ocount != {3}:
            bad.append((persistent, 'step counts', sorted(tcount), sorted(ocount)))
        tr._grad_sync = None
    q.put(('ok', bad, 3))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fused_adopt_with_per_rank_text_drop(emu_lib):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_adopt_worker, args=(r, 2, port, str(emu_lib), q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _queue
    res, waited = [], 0
    while len(res) < len(procs) and waited < 900:
        try:
            res.append(q.get(timeout=5))
        except _queue.Empty:
            waited += 5
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead:
                for p in procs:
                    p.kill()
                raise AssertionError(f'a rank exited with {dead} (see its traceback above)')
    assert len(res) == len(procs), 'timed out'
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for status, bad, _ in res:
        assert status == 'ok' and not bad, bad[:10]
