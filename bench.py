"""bench.py -- one data-parallel training step (forward + backward [+ RCCL gradient all-reduce]) of the E2-TTS
flow-matching transformer on synthetic data, BASELINE.json's metric on BASELINE.json's config.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is measured live: every launch of the dominant kernel
(e2k_gemm_nt_bf16) inside the timed region is bracketed with HIP events on its own stream.
`cpu_baseline` times the CPU oracle (kind "port": the reference itself cannot be imported)
on a bounded sample of the same workload, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import math
import json
import os
import random
import sys
import time
from pathlib import Path

# The step runs on up to five HIP streams (the caller's, the TEXT and WGRAD launch lanes, the gradient-exchange side stream,
# RCCL's own).  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); a fifth stream shares a
# queue with a lane and serialises against it: measured on MI355X, merely creating an RCCL communicator took the cfg3 step
# from 93.3 to 104.3 ms, and GPU_MAX_HW_QUEUES=8 brought it back to 93.3 (profiles/r03_hw_queues.jsonl).  Must be set
# before the HIP runtime initialises, i.e. before the first device call of this process.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
# Kernel arguments in device memory instead of host-coherent memory: the command processor fetches them without crossing the bus, which
# shortens the gap between dependent launches -- and a step is ~1600 launches in three dependent chains.  MI355X, same box, interleaved:
# cfg3 86.53 / 86.45 -> 85.51 / 85.59 ms, cfg2 14.92 / 14.95 -> 14.44 / 14.44 ms (profiles/r06g_kernarg_ab.txt).  Read by the HIP runtime at
# its initialisation, like the variable above.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

ROOT = Path(__file__).resolve().parent
for p in (ROOT / 'e2-tts-pytorch_amd', ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # name: (dim, depth, heads, B per GPU, T)
    'cfg2': (512, 8, 8, 8, 1024),
    'cfg3': (1024, 24, 16, 8, 1024),
}
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def step_flops(dim, depth, heads, B, T, text_on=True):
    """algorithmic dense FLOPs of one fwd+bwd step"""
    D, L, N = dim, depth, T + 32
    I, Dt, m, s = heads * 64, dim // 2, 4, 4
    f_tok = L * (8 * D * I + 4 * N * I + 6 * m * D * D) + 2 * s * L * D * D
    if text_on:
        f_tok += L * (8 * Dt * I + 4 * N * I + 6 * m * Dt * Dt) + s * (L * 2 * (D + Dt) * D + (L - 1) * 2 * (D + Dt) * Dt)
    f_io = 2 * (2 * 100 * D) + 2 * D * 100
    return 3 * B * (N * f_tok + T * f_io)


def _groups(rows, nprof):
    """per-step HIP-event milliseconds, launches and achieved TFLOP/s of every e2k call name in a profiled plan replay"""
    g = {}
    for r in rows:
        e = g.setdefault(r['name'], [0.0, 0, 0.0])
        e[0] += r['ms']
        e[1] += 1
        e[2] += r['flops']
    out = {}
    for k, (ms, n, fl) in sorted(g.items(), key=lambda kv: -kv[1][0]):
        out[k] = {'ms': round(ms / nprof, 3), 'launches': n // nprof}
        if fl > 0:
            out[k]['tflops'] = round(fl / (ms * 1e-3) / 1e12, 1)
    return out


def _lane_ms(rows, nprof):
    """per-step milliseconds of the calls of each launch lane (profiled replay: every call alone on one stream), fwd / bwd"""
    try:
        if not rows:
            return None
        names = ('main', 'text', 'wgrad', 'lane3')
        return {names[k]: {ph: round(sum(r['ms'] for r in rows if r.get('lane', 0) == k and r['phase'] == ph) / nprof, 2) for ph in ('fwd', 'bwd')}
                for k in sorted({r.get('lane', 0) for r in rows})}
    except Exception as e:      # noqa: BLE001  (a diagnostic must not cost the bench line)
        return {'error': repr(e)}


def host_launch_floor(depth, dev, dropout, graphs=None):
    """host time per training step when the GPU is NOT the bottleneck: the same depth (= the same number of recorded
    launches per step) at a tiny width / batch / length, so that the kernels are negligible.  `host_enqueue_ms_per_step`
    of the real workload cannot show this: once the HIP queue is full the enqueuing thread blocks until the GPU frees a
    slot, so on a GPU-bound step it simply tracks the GPU time."""
    from e2_tts_pytorch_amd import E2TTS
    m = E2TTS(transformer=dict(dim=256, depth=depth, heads=4, dropout=dropout), use_vocos=False, cond_drop_prob=0.).to(dev).train()
    tr = m.transformer
    tr.enable_persistent_grads()
    if graphs is not None:
        tr.enable_graphs(graphs)
    flat = {id(q) for q, _ in tr._layout.slots}
    rest = [p for p in m.parameters() if id(p) not in flat]
    mel, text = torch.randn(1, 96, 100, device=dev), ['launch floor']

    def step():
        m(mel, text=text).loss.backward()
        for p in rest:
            p.grad = None
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 8
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    return {'host_ms_per_step': t_host / n * 1e3, 'wall_ms_per_step': t_all / n * 1e3, 'graphs': bool(tr._graphs_on),
            'probe': f'E2TTS(dim=256, depth={depth}, heads=4), B=1, T=96: same launches per step, negligible kernel time'}


def synthetic_text(B, seed):
    rng = random.Random(seed)
    alphabet = ''.join(chr(c) for c in range(32, 127))
    return [''.join(rng.choice(alphabet) for _ in range(rng.randint(20, 200))) for _ in range(B)]


def cpu_baseline(dim, depth, heads, T, sample_depth=None, threads=None):
    """oracle (fp32, torch CPU) on a bounded sample of the same workload: the FULL model (same width / heads / depth / sequence
    length) at B = 1, one fwd+bwd (about 26 s on the GPU box's host: no extrapolation over layers; `sample_depth` < depth times
    fewer layers and scales, for quick runs).  Thread count capped at 32: torch's eager CPU ops are bandwidth bound and slow down
    badly when a 256-thread host oversubscribes them."""
    from oracle import e2tts_oracle as O
    threads = threads or min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    random.seed(0)
    torch.manual_seed(0)
    sd = min(sample_depth or depth, depth)
    model = O.E2TTS(transformer=dict(dim=dim, depth=sd, heads=heads), cond_drop_prob=0.)
    text = synthetic_text(1, 1)
    warm = model(torch.randn(1, 64, 100), text=text)          # thread pool / allocator warm-up, not timed
    warm.loss.backward()
    model.zero_grad()
    mel = torch.randn(1, T, 100)
    t0 = time.perf_counter()
    out = model(mel, text=text)
    out.loss.backward()
    dt = time.perf_counter() - t0
    return {'value': T / dt * sd / depth, 'unit': 'mel-frames/s', 'cores': threads, 'kind': 'port',
            'sample': f'CPU oracle (fp32 torch eager, {threads} threads): dim={dim} heads={heads} T={T} B=1, '
                      f'{sd} of {depth} layers, one fwd+bwd = {dt:.1f} s measured' + ('' if sd == depth else f'; value = T/dt scaled by {sd}/{depth} to the full-depth step') +
                      f'.  The port reproduces the reference source bit for bit where that can be '
                      f'executed (oracle/pin_against_reference.py); /root/reference itself cannot travel to the GPU box'}


def optimizer_leg(model, net, mel, text, noise, ms_plain, k=6):
    """what the trainer adds to a step (trainer.py:270-279): global-norm clip + ADOPT over the flat buffers (optim.FusedAdopt: one sumsq + one
    update launch for the backbone) and the EMA update (optim.FusedEMA; every 10th step in the reference configuration, ema_pytorch's
    default), on its own and folded into the ADOPT pass (FusedAdopt.attach_ema).  Same plan, same inputs as the headline loop, lr tiny so
    that the steps leave the model alone.

    ONE loop whose steps cycle through the four modes, a HIP event at the start of every step: a step's time is its start event to the
    next one (the host runs ahead, the queue never drains), and each mode's cost is the difference of MEDIANS against the plain steps of
    the same loop.  (Rounds 1-4 timed three loops one after the other and charged the optimizer with the warming of the chip between
    them -- a cfg3 step goes from ~84 to ~87 ms over the first ten seconds of load: 6.9 + 2.5 ms reported where the launches themselves
    took 4.6 + 1.7, profiles/r05l_optimizer_leg_in_context.json.)"""
    from e2_tts_pytorch_amd.optim import FusedAdopt, FusedEMA
    opt = FusedAdopt(model, lr=1e-7, max_grad_norm=1.0)
    ema = FusedEMA(model, update_after_step=0, update_every=1)
    modes = [('fwd_bwd', False, False, False), ('with_clip_adopt', True, False, False), ('with_clip_adopt_ema', True, True, False),
             ('with_clip_adopt_ema_folded', True, True, True)]
    ev, tags = [], []
    for i in range(4 * (k + 1) + 1):
        name, with_opt, with_ema, fold = modes[i % 4]
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append(e)
        tags.append(name)
        out = net(mel, text=text, _noise=noise)
        out.loss.backward()
        opt.attach_ema(ema if fold else None)
        if with_opt:
            opt.step()
        opt.zero_grad(set_to_none=True)
        if with_ema:
            ema.update()
    torch.cuda.synchronize()
    out = {}
    for name, *_ in modes:
        d = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(4, len(ev) - 1) if tags[i] == name)      # (the first cycle is warm-up)
        out['ms_per_step_' + name] = d[len(d) // 2]
    out['ms_per_step_fwd_bwd_headline'] = ms_plain
    out['clip_adopt_ms'] = out['ms_per_step_with_clip_adopt'] - out['ms_per_step_fwd_bwd']
    out['ema_update_ms'] = out['ms_per_step_with_clip_adopt_ema_folded'] - out['ms_per_step_with_clip_adopt']
    out['ema_update_separate_launch_ms'] = out['ms_per_step_with_clip_adopt_ema'] - out['ms_per_step_with_clip_adopt']
    out['steps_per_mode'] = k
    out['note'] = ('fused clip + ADOPT (per-parameter steps, text-stream group skipped on text-dropped steps) and EMA over the flat fp32 '
                   'buffers, trainer.py:270-279; ema_update_ms = the EMA folded into the ADOPT pass (FusedAdopt.attach_ema), '
                   'ema_update_separate_launch_ms = FusedEMA.update() as a pass of its own; the EMA update runs every 10th step in the '
                   'reference configuration.  Steps of the four modes are interleaved in one loop and timed by HIP events at the step '
                   'boundaries; differences of medians against the plain steps of the same loop (ms_per_step_fwd_bwd here is a step of a '
                   'chip that has been under load for several seconds, the headline ms_per_step the first seconds)')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='cfg3', choices=sorted(CONFIGS))
    ap.add_argument('--dropout', type=float, default=0.1, help='reference training default (e2_tts.py:540)')
    ap.add_argument('--drop-text', action='store_true', help='run with the text stream dropped (CFG null pass cost)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-launch-floor', action='store_true', help='skip the host_launch_floor probe (a tiny model with the same launch '
                    'count per step: its kernels would dilute a rocprofv3 per-kernel summary of this command)')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--eager', action='store_true', help='drive every kernel launch from Python instead of replaying the recorded '
                    'launch plan from C++ (A/B: ~35 us of host time per launch, the step becomes host bound)')
    ap.add_argument('--autograd-grads', action='store_true',
                    help='hand the backbone gradients to autograd every step (default here: Transformer.enable_persistent_grads(), one '
                         'flat gradient buffer that each backward overwrites -- what an optimizer over the flat buffer consumes)')
    ap.add_argument('--force-ddp', action='store_true', help='wrap in ddp.DataParallel even with one rank (exercises the RCCL path)')
    ap.add_argument('--grad-dtype', default='bf16', choices=['fp32', 'bf16'],
                    help='element type of the gradient slabs on the xGMI links (bf16 halves the bytes: 1.45 instead of 2.9 GB per step)')
    ap.add_argument('--bucket-layers', type=int, default=1, help='layer slabs merged per all-reduce')
    ap.add_argument('--ddp-bisect', default=None, choices=['init', 'nohook', 'nooutside'],
                    help='diagnosis of the --force-ddp overhead: init = only init_process_group (no wrapper); nohook = wrapper without the slab hook; nooutside = wrapper without the non-backbone all-reduce')
    ap.add_argument('--ddp-defer', action='store_true', help='one gradient all-reduce after the backward pass instead of per-layer slabs overlapped with it (A/B)')
    ap.add_argument('--no-optimizer-leg', action='store_true', help='skip the extra leg that times the step WITH the fused gradient clip + '
                    'ADOPT update (+ EMA) after the headline measurement (N = 1 only; it never enters `value`)')
    ap.add_argument('--graphs', type=int, default=None, choices=[0, 1], help='replay the recorded plans as HIP graphs (Transformer.enable_graphs; '
                    'default: the package default / E2K_GRAPH)')
    ap.add_argument('--no-warm-leg', action='store_true', help='skip ms_per_step_warm (>= --warm-seconds of load, then K event-timed steps)')
    ap.add_argument('--warm-seconds', type=float, default=10.0)
    ap.add_argument('--main-cus', default=None, help='first:count -- run the step on a HIP stream confined to these CUs '
                    '(hipExtStreamCreateWithCUMask; A/B of the launch lanes with E2K_LANE_CUS, DESIGN.md section 5.1)')
    ap.add_argument('--dump-ops', default=None, help='write the per-shape launch table of the profiled plan replays (name, flops, count, '
                    'average ms, TFLOP/s) to this JSON file')
    args = ap.parse_args()

    from e2_tts_pytorch_amd import E2TTS, ops
    from e2_tts_pytorch_amd.ddp import DataParallel

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1 or args.force_ddp or args.ddp_bisect:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if os.environ.get('E2K_BENCH_LAZY_PG') == '1':          # (diagnosis: communicator created at the first collective instead of here)
            dist.init_process_group('nccl')
        else:
            dist.init_process_group('nccl', device_id=dev)

    dim, depth, heads, B, T = CONFIGS[args.config]
    B = args.batch or B
    random.seed(1234)                     # same python RNG on every rank: identical init + identical CFG coin flips
    torch.manual_seed(1234)
    model = E2TTS(transformer=dict(dim=dim, depth=depth, heads=heads, dropout=args.dropout), use_vocos=False,
                  cond_drop_prob=0.).to(dev)
    model.train()
    net = DataParallel(model, grad_dtype=torch.bfloat16 if args.grad_dtype == 'bf16' else torch.float32,
                       bucket_layers=args.bucket_layers, defer=args.ddp_defer) if (world > 1 or args.force_ddp or args.ddp_bisect in ('nohook', 'nooutside')) else model
    if args.ddp_bisect == 'nohook':
        for bb in net._backbones:
            bb._grad_sync = None
    if args.ddp_bisect == 'nooutside':
        net._outside = []
    torch.manual_seed(1000 + rank)        # different synthetic data per rank (weak scaling: B per GPU fixed)
    mel = torch.randn(B, T, 100, device=dev)
    text = synthetic_text(B, 1000 + rank)
    noise = {'drop_text_cond': True} if args.drop_text else None

    tr = model.transformer
    tr.enable_plans(not args.eager)
    if args.graphs is not None:
        tr.enable_graphs(bool(args.graphs))
    params = list(model.parameters())
    if not args.autograd_grads:
        tr.enable_persistent_grads()
        flat = {id(q) for q, _ in tr._layout.slots}
        params = [p for p in params if id(p) not in flat]          # (the backbone's gradients are overwritten in place)

    def step():
        out = net(mel, text=text, _noise=noise)
        out.loss.backward()
        for p in params:                  # == optimizer.zero_grad(set_to_none=True) (trainer.py:277) without walking the
            p.grad = None                 #    module tree every step (model.zero_grad costs ~10 ms of host time here)
        return out.loss

    if args.main_cus:
        a, n = (int(v) for v in args.main_cus.split(':'))
        torch.cuda.set_stream(ops.cu_masked_stream(dev, a, n))

    graphs_on = bool(getattr(tr, '_graphs_on', False)) and not args.eager
    launch_mode_note = 'eager launches from Python (--eager)' if args.eager else (
        'recorded launch plan replayed as ONE HIP graph per pass (e2k_plan_graph_launch; with a gradient exchange the backward pass stays on '
        'the segmented eager replay)' if graphs_on else
        'recorded launch plan re-issued from C++ (e2k_plan_run; eager stream launches, no HIP graph)')
    # a signature is recorded the second time it is seen (forward) and at its first backward; do that outside the W
    # warm-up steps so that even --warmup 0 times replayed steps only
    step()
    step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enqueue = time.perf_counter() - t0          # host time to enqueue everything (no sync inside the loop)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss_val = float(loss.item())

    # roofline leg: HIP events on the launch stream around every recorded launch of the step (e2k_plan_profile replays
    # the very plan the timed region ran, one event after each call); the committed rocprofv3 summary (profiles/) must
    # agree with these averages.  With --eager the NT launches are bracketed by torch events instead.
    prof_rows, nprof = [], 2
    lane_ops = 0
    if not args.eager:
        for _ in range(nprof):
            prof_rows += tr.plan_profile()
        # (the profiled replay runs on ONE stream: the lane ordering points are skipped there and drop out of the tables;
        # the per-call times are those of each kernel running alone, their sum exceeds the step time when lanes overlap)
        lane_ops = sum(r['name'].startswith('lane_event') for r in prof_rows) // nprof
        prof_rows = [r for r in prof_rows if not r['name'].startswith('lane_event')]
        gemm = [r for r in prof_rows if r['name'] in ('gemm_nt_bf16', 'gemm_nt_geglu_bf16', 'gemm_nt_geglu_bwd_bf16', 'gemm_nt_qkrot_bf16')]     # (the NT kernel family: plain, + GEGLU forward epilogue (inference), + GEGLU backward epilogue, + rotary q / k epilogue)
        gemm_flops, gemm_ms, n_launch = sum(r['flops'] for r in gemm), sum(r['ms'] for r in gemm), len(gemm)
    else:
        prof = []
        ops.set_gemm_profile(prof)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        ops.set_gemm_profile(None)
        gemm_flops = sum(f for f, _, _ in prof)
        gemm_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in prof)
        n_launch = len(prof)

    if args.dump_ops and prof_rows and rank == 0:
        by = {}
        for r in prof_rows:
            k = (r['name'], int(r['flops']))
            e = by.setdefault(k, [0, 0.0])
            e[0] += 1
            e[1] += r['ms']
        table = [dict(name=k[0], flops=k[1], launches_per_step=c // nprof, avg_ms=round(m / c, 5), ms_per_step=round(m / nprof, 4),
                      tflops=(round(k[1] / (m / c) / 1e9, 1) if k[1] else None)) for k, (c, m) in by.items()]
        table.sort(key=lambda e: -e['ms_per_step'])
        Path(args.dump_ops).parent.mkdir(parents=True, exist_ok=True)
        json.dump(table, open(args.dump_ops, 'w'), indent=1)

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # warm leg: the contract's W warm-up steps + K timed steps see a chip in its first seconds of load (83-84 ms at cfg3 in round 5); a
    # training job lives at the clocks the chip settles to after ~10 s (86-87 ms).  Same plan, same step(): >= 10 s of continuous load,
    # then K steps timed by HIP events at the step boundaries, median (max over ranks).  Reported beside the headline, never as `value`.
    ms_warm, warm_load_s = None, None
    if not args.no_warm_leg:
        n_load = int(math.ceil(args.warm_seconds / max(dt / args.steps, 1e-4)))           # (dt is the max over ranks: every rank runs the same count)
        tw = time.perf_counter()
        for _ in range(n_load):
            step()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        evs[0].record()
        for i in range(args.steps):
            step()
            evs[i + 1].record()
        torch.cuda.synchronize()
        warm_load_s = time.perf_counter() - tw
        d = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
        tw_ = torch.tensor([d[len(d) // 2]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tw_, op=dist.ReduceOp.MAX)
        ms_warm = float(tw_.item())
    if rank == 0:
        ms = dt / args.steps * 1e3
        frames = B * T * world
        sf = step_flops(dim, depth, heads, B, T, text_on=not args.drop_text)
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # HBM traffic of the dominant kernel per launch: PMC counters cannot be read from inside this process, so the
        # number comes from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same launch mix
        # (profiles/r06_nt_traffic.json, produced by tools/nt_shapes.py + tools/nt_traffic_probe.py + tools/nt_traffic_reduce.py); cfg3 only
        traffic, traffic_note = None, None
        tf = ROOT / 'profiles' / 'r06_nt_traffic.json'
        if not tf.exists():
            tf = ROOT / 'profiles' / 'r05_nt_traffic.json'
        if args.config == 'cfg3' and B == CONFIGS['cfg3'][3] and not args.drop_text and tf.exists():
            tj = json.load(open(tf))
            traffic = tj['traffic_bytes_per_launch']
            traffic_note = (f"bytes per launch from {tf.name}: fetch {tj['hbm_fetch_bytes_per_launch'] / 1e6:.1f} MB (FETCH_SIZE x2, gfx950) + "
                            f"write {tj['hbm_write_bytes_per_launch'] / 1e6:.1f} MB; algorithmic {tj['algorithmic_bytes_per_launch'] / 1e6:.1f} MB "
                            "(A + B + C + residual)")
        res = {
            'metric': 'mel-frames/sec (fwd+bwd training step)',
            'value': frames / (dt / args.steps),
            'unit': 'mel-frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'ms_per_step_warm': ms_warm,
            'warm_note': (f'median of {args.steps} steps (HIP events at the step boundaries, max over ranks) after {warm_load_s:.1f} s of continuous load '
                          f'of the same replayed plan: the clocks a training job runs at; `value` / ms_per_step are the contract\'s W warm-up + K steps, '
                          f'i.e. the first seconds of load') if ms_warm is not None else None,
            'mfma_roofline_frac_whole_step_warm': (sf / (ms_warm * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if ms_warm else None,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {
                'workload': f'{args.config}: E2TTS(dim={dim}, depth={depth}, heads={heads}) fwd+bwd, B={B}/GPU, T={T}, '
                            f'n_mels=100, random-init weights, {"text stream dropped" if args.drop_text else "text stream on (cond_drop_prob=0)"}, '
                            f'dropout={args.dropout}, bf16 MFMA compute / fp32 master weights+grads',
                'global_batch': B * world, 'seq_len': T, 'parallelism': f'dp{world}',
                'grad_exchange': ((f'{args.grad_dtype}, ONE all-reduce after the backward pass' if args.ddp_defer else f'{args.grad_dtype} slabs, {args.bucket_layers} layer(s) per all-reduce, side stream') if (world > 1 or args.force_ddp) else None),
            },
            'step_tflops_algorithmic': sf / 1e12,
            'model_tflops_per_s_per_gpu': sf / (dt / args.steps) / 1e12,
            'mfma_roofline_frac_whole_step': sf / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS,
            'loss': loss_val,
            'host_enqueue_ms_per_step': t_enqueue / args.steps * 1e3,
            'cu_masks': {'main': args.main_cus, 'lanes': os.environ.get('E2K_LANE_CUS')},
            'launch_mode': launch_mode_note, 'gemm_flags': ops.gemm_flags, 'fuse_geglu': bool(ops.fuse_geglu),
            'launches_per_step': (len(prof_rows) // nprof) if prof_rows else None,
            'lane_ms_per_step': _lane_ms(prof_rows, nprof),          # work of each lane, every call timed alone: the step cannot be shorter than the longest chain
            'launch_lanes': {'on': bool(getattr(tr, '_lanes_on', False)), 'backward': bool(getattr(tr, '_lanes_bwd', False)), 'ordering_points_per_step': lane_ops,
                             'note': 'text branches / weight-gradient GEMMs on side streams (ops.Lanes, csrc/plan.h); E2K_LANES=0 for the single-stream schedule'},
            'kernel_groups_ms_per_step': _groups(prof_rows, nprof) if prof_rows else None,
            'roofline': {
                'bound': 'mfma', 'kernel': 'e2k_gemm_nt_bf16 (+ e2k_gemm_nt_geglu_bwd_bf16 / e2k_gemm_nt_qkrot_bf16, the same kernel with the GEGLU backward / the rotary embedding of q and k as its epilogue): gemm_nt_256_kernel (256x256x64, 8-phase) / gemm_nt_glds_kernel (128x128x64), C tiles through LDS in whole-line 16-byte stores (bf16 MFMA 16x16x32; every forward and dgrad GEMM of the step; no remainder split since round 6: alone-timed, as here, that costs the kernel 2 % -- 0.280 -> 0.270 of peak on one box -- and gains the step 0.5 ms, profiles/r06f_nt_remainder_split_in_step_ab.txt)',
                'achieved': achieved, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_BF16_TFLOPS,
                'launches_per_step': n_launch / nprof,
                'avg_launch_ms': gemm_ms / max(n_launch, 1),
                'flops_per_launch_avg': gemm_flops / max(n_launch, 1),
                'time_share_of_step': (gemm_ms / nprof) / ms,
                'measured': 'HIP events on the launch stream around every recorded launch (e2k_plan_profile), 2 replays of the '
                            'timed plan right after the timed region, every call ALONE on one stream; the rocprofv3 summary that '
                            'agrees with avg_launch_ms is the single-stream one (E2K_LANES=0, profiles/r06_bench_cfg3_kernel_stats_single_stream.csv): '
                            'with the launch lanes the kernels of different lanes overlap and stretch (profiles/r06_bench_cfg3_kernel_stats_lanes.csv).  Since the end of '
                            'round 3 outputs of 64-223 tiles of 256 x 256 (the 8448-token GEMMs with N <= 2048) run the 256 x 256 kernel on '
                            'a part of the CUs: in the step the other launch lanes use the rest (step -1.5 to -3 %, '
                            'profiles/r03_t256_threshold_ab.jsonl); timed ALONE, as here, the same choice costs 4 % (751 -> 724 TFLOP/s)',
                'traffic': traffic, 'traffic_note': traffic_note,
            },
        }
        if world == 1 and not args.eager and not args.no_launch_floor:      # (rank 0 alone would keep the other ranks waiting)
            try:
                res['host_launch_floor'] = host_launch_floor(depth, dev, args.dropout, graphs=graphs_on)
                res['host_launch_floor_other_mode'] = host_launch_floor(depth, dev, args.dropout, graphs=not graphs_on)
            except Exception as e:      # noqa: BLE001
                res['host_launch_floor'] = {'error': repr(e)}
        if world == 1 and not args.eager and not args.no_optimizer_leg:
            # The headline metric is BASELINE.json's fwd + bwd; what the trainer adds per step (trainer.py:270-279: clip_grad_norm_,
            # Adopt.step, zero_grad, EMA.update) is timed here on the same model, after the timed region, and reported beside it
            try:
                res['optimizer_leg'] = optimizer_leg(model, net, mel, text, noise, ms)
            except Exception as e:      # noqa: BLE001
                res['optimizer_leg'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res['cpu_baseline'] = cpu_baseline(dim, depth, heads, T)
            except Exception as e:      # noqa: BLE001
                res['cpu_baseline'] = {'value': None, 'unit': 'mel-frames/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': f'failed: {e!r}'}
        print(json.dumps(res), flush=True)
    if world > 1 or args.force_ddp:
        dist.barrier()                    # nobody tears the communicator down while rank 0 is still reporting
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
