"""Where do the milliseconds between `fwd + bwd` and `fwd + bwd + clip + ADOPT` go?  bench.py's optimizer leg reports 7.3 ms for
4.05 ms of optimizer kernels (rocprofv3 averages).  This traces two consecutive training steps with torch.profiler and prints
the device timeline around the optimizer kernels: gaps between consecutive device events, and host time of opt.step().
-> gpurun_out/optimizer_leg_trace.json"""
import json, os, random, sys, time
from pathlib import Path
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synthetic_text
from e2_tts_pytorch_amd import E2TTS
from e2_tts_pytorch_amd.optim import FusedAdopt
dev = torch.device('cuda')
random.seed(1234); torch.manual_seed(1234)
dim, depth, heads, B, T = (int(a) for a in (sys.argv[1:6] or (1024, 24, 16, 8, 1024)))
model = E2TTS(transformer=dict(dim=dim, depth=depth, heads=heads, dropout=0.1), use_vocos=False, cond_drop_prob=0.).to(dev).train()
tr = model.transformer
tr.enable_persistent_grads()
mel = torch.randn(B, T, 100, device=dev)
text = synthetic_text(B, 1000)
opt = FusedAdopt(model, lr=1e-7, max_grad_norm=1.0)
host = []
def train_step():
    out = model(mel, text=text)
    out.loss.backward()
    t0 = time.perf_counter()
    opt.step()
    opt.zero_grad(set_to_none=True)
    host.append((time.perf_counter() - t0) * 1e3)
for _ in range(5):
    train_step()
torch.cuda.synchronize()
host.clear()
t0 = time.perf_counter()
for _ in range(4):
    train_step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 4 * 1e3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    train_step()
    train_step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
rows = [dict(name=e.name[:70], start_us=e.time_range.start, dur_us=e.time_range.end - e.time_range.start) for e in evs]
t_first = rows[0]['start_us']
# device busy intervals (union over streams) and the idle gaps between them
busy, gaps, cur_end = 0.0, [], rows[0]['start_us']
for r in rows:
    s, e = r['start_us'], r['start_us'] + r['dur_us']
    if s > cur_end:
        gaps.append((s - cur_end, r['name'], s - t_first))
        busy += e - s
    else:
        busy += max(0.0, e - cur_end)
    cur_end = max(cur_end, e)
span = cur_end - t_first
opt_idx = [i for i, r in enumerate(rows) if 'adopt_kernel' in r['name'] or 'sumsq_kernel' in r['name']]
print(f'wall per train step {wall:.2f} ms; host time of opt.step() + zero_grad {sum(host) / len(host):.2f} ms')
print(f'traced span (2 steps) {span / 1e3:.2f} ms, device busy (union) {busy / 1e3:.2f} ms, idle {(span - busy) / 1e3:.2f} ms in {len(gaps)} gaps')
gaps.sort(reverse=True)
print('largest idle gaps (us, the event that ended them, offset ms):')
for g, n, off in gaps[:15]:
    print(f'  {g:9.1f}  {n[:60]:60s} at {off / 1e3:8.2f}')
if opt_idx:
    a, b = max(0, opt_idx[0] - 6), min(len(rows), opt_idx[0] + 26)
    print('timeline around the first optimizer kernels (offset ms, duration us, name):')
    for r in rows[a:b]:
        print(f"  {(r['start_us'] - t_first) / 1e3:9.3f} {r['dur_us']:9.1f}  {r['name']}")
# the same kernels with and without the optimizer at the end of every step: device time per kernel name over 3 steps each
def trace_totals(fn, n=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as pr:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    tot, seq = {}, []
    for e in pr.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            d = e.time_range.end - e.time_range.start
            tot[e.name[:60]] = tot.get(e.name[:60], 0.0) + d / n
            seq.append((e.time_range.start, e.name[:60], d))
    return tot, sorted(seq)

def plain_step():
    out = model(mel, text=text)
    out.loss.backward()
    opt.zero_grad(set_to_none=True)

tp, seq_p = trace_totals(plain_step)
tt, seq_t = trace_totals(train_step)


def boundary(seq, label):
    # union busy time, span per step, and the kernels around the boundary between the first and the second traced step
    t0, cur, busy = seq[0][0], seq[0][0], 0.0
    for s_, _, d in seq:
        e_ = s_ + d
        if s_ > cur:
            busy += d
        else:
            busy += max(0.0, e_ - cur)
        cur = max(cur, e_)
    print(f'{label}: span of 3 steps {(cur - t0) / 1e3:.2f} ms = {(cur - t0) / 3e3:.2f} per step, device busy (union) {busy / 3e3:.2f} ms per step')
    idx = [i for i, (_, n, _) in enumerate(seq) if 'stream_pack_fwd' in n]
    if len(idx) >= 4:
        i0 = idx[2]            # first stream_pack_fwd of the second step (audio; the text one follows)
        print(f'  kernels around the start of the second step ({label}); offset ms, duration us, name')
        for s_, n, d in seq[max(0, i0 - 40):i0 + 4]:
            print(f'    {(s_ - t0) / 1e3:9.3f} {d:8.1f}  {n}')


boundary(seq_p, 'plain loop')
boundary(seq_t, 'with optimizer')
names = sorted(set(tp) | set(tt), key=lambda k: -(tt.get(k, 0) - tp.get(k, 0)))
print(f'device time per step (sum over kernels): plain {sum(tp.values()) / 1e3:.2f} ms, with optimizer {sum(tt.values()) / 1e3:.2f} ms')
print('largest per-kernel differences (us per step: with optimizer - plain, plain, ratio):')
diffs = []
for k in names[:14] + names[-4:]:
    a, b = tp.get(k, 0.0), tt.get(k, 0.0)
    print(f'  {b - a:9.1f} {a:10.1f} {b / a if a else float("nan"):6.3f}  {k}')
    diffs.append(dict(kernel=k, plain_us=a, with_optimizer_us=b))
(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(dict(per_kernel=diffs, plain_device_ms=sum(tp.values()) / 1e3, with_optimizer_device_ms=sum(tt.values()) / 1e3,
               wall_ms_per_train_step=wall, host_ms_opt_step=sum(host) / len(host), span_ms=span / 1e3, busy_ms=busy / 1e3,
               largest_gaps=[dict(us=g, before=n, at_ms=off / 1e3) for g, n, off in gaps[:30]]),
          open(ROOT / 'gpurun_out' / 'optimizer_leg_trace.json', 'w'), indent=1)
