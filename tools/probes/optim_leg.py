"""where the optimizer leg of bench.py goes: the bench's cfg3 training loop with FusedAdopt (+ FusedEMA, folded or not), HIP events
around opt.step() / ema.update() inside the running loop (GPU time of exactly those launches, in context) and host time per phase.
    python tools/probes/optim_leg.py [out.json]"""
import json, random, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
import bench
from e2_tts_pytorch_amd import E2TTS
from e2_tts_pytorch_amd.optim import FusedAdopt, FusedEMA

dim, depth, heads, B, T = bench.CONFIGS['cfg3']
dev = torch.device('cuda', 0)
random.seed(1234); torch.manual_seed(1234)
model = E2TTS(transformer=dict(dim=dim, depth=depth, heads=heads, dropout=0.1), use_vocos=False, cond_drop_prob=0.).to(dev).train()
torch.manual_seed(1000)
mel = torch.randn(B, T, 100, device=dev)
text = bench.synthetic_text(B, 1000)
tr = model.transformer
tr.enable_plans(True)
tr.enable_persistent_grads()
opt = FusedAdopt(model, lr=1e-7, max_grad_norm=1.0)
ema = FusedEMA(model, update_after_step=0, update_every=1)
res = {}


def loop(name, with_opt, with_ema, k=8):
    host = dict(fwd=0., bwd=0., opt=0., zero=0., ema=0.)
    ev = []

    def one(timed):
        t = [time.perf_counter()]
        out = model(mel, text=text); t.append(time.perf_counter())
        out.loss.backward(); t.append(time.perf_counter())
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        if with_opt:
            opt.step()
        e[1].record(); t.append(time.perf_counter())
        opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
        if with_ema:
            ema.update()
        e[2].record(); t.append(time.perf_counter())
        if timed:
            ev.append(e)
            for key, a, b in zip(host, t[:-1], t[1:]):
                host[key] += (b - a) * 1e3 / k
    for _ in range(3):
        one(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        one(True)
    t_host = (time.perf_counter() - t0) / k * 1e3
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / k * 1e3
    res[name] = dict(ms_per_step=ms, host_enqueue_ms=t_host, host_phase_ms=host,
                     gpu_opt_segment_ms=sorted(e[0].elapsed_time(e[1]) for e in ev)[k // 2],
                     gpu_ema_segment_ms=sorted(e[1].elapsed_time(e[2]) for e in ev)[k // 2])
    print(name, json.dumps(res[name]), flush=True)


loop('fwd_bwd', False, False)
loop('clip_adopt', True, False)
loop('clip_adopt_ema', True, True)
opt.attach_ema(ema)
loop('clip_adopt_ema_folded', True, True)
loop('fwd_bwd_again', False, False)

# the same question without the drift of a warming chip between loops: ONE loop whose steps cycle through the four modes, an event at
# the start of every step; a step's GPU time = its start event to the next one (the queue never drains: the host runs ahead)
def cycle(k=40):
    modes = [('plain', False, False, False), ('clip_adopt', True, False, False), ('clip_adopt_ema', True, True, False),
             ('clip_adopt_ema_folded', True, True, True)]
    ev, tags = [], []
    for i in range(k + 4):
        name, with_opt, with_ema, fold = modes[i % 4]
        e = torch.cuda.Event(enable_timing=True); e.record()
        ev.append(e); tags.append(name)
        model(mel, text=text).loss.backward()
        opt._ema = ema if fold else None
        if with_opt:
            opt.step()
        opt.zero_grad(set_to_none=True)
        if with_ema:
            ema.update()
        else:
            ema._folded = None
    torch.cuda.synchronize()
    out = {}
    for name, *_ in modes:
        d = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(4, k + 3) if tags[i] == name)
        out[name] = dict(median_ms=d[len(d) // 2], min_ms=d[0], max_ms=d[-1], n=len(d))
    for name in ('clip_adopt', 'clip_adopt_ema', 'clip_adopt_ema_folded'):
        out[name]['over_plain_ms'] = out[name]['median_ms'] - out['plain']['median_ms']
    return out


ema._folded = None
res['interleaved'] = cycle()
print('interleaved', json.dumps(res['interleaved']), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)
