"""the optimizer-side kernels alone at cfg3 size (716 M fp32 elements per buffer), HIP events, one process per variant (the switches are
read once per process): E2K_OPTIM_VARIANT bits 1 = contiguous spans, 2 = temporal accesses, 4 = correctly rounded sqrt / divide;
E2K_OPTIM_GRID = grid cap.   python tools/probes/optim_ab.py [out.json]"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
N = 716_258_020 // 4 * 4

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
    import torch
    from e2_tts_pytorch_amd import ops
    dev = 'cuda'
    torch.manual_seed(0)
    p, g, m, e = (torch.randn(N, device=dev) * s for s in (1., 1e-3, 0.1, 1.))
    v = torch.rand(N, device=dev) * 1e-6 + 1e-9
    starts = torch.arange(48, dtype=torch.int64) * (N // 48 // 4 * 4) + 1024 * 1024
    ranges = torch.stack([starts, starts + 4 * 1024 * 1024], 1).to(torch.int32).to(dev)
    gs = torch.zeros(1, dtype=torch.float64, device=dev)

    def timed(fn, k=5):
        fn(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        ev[0].record()
        for i in range(k):
            fn(); ev[i + 1].record()
        torch.cuda.synchronize()
        return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(k))[k // 2]

    kw = dict(lr=1e-7, max_grad_norm=1.0, gsumsq=gs, ranges=ranges, step_b=3)
    out = dict(
        sumsq_ms=timed(lambda: ops.sumsq(g, gs)),
        adopt_ms=timed(lambda: ops.adopt_step(p, g, m, v, 3, active_b=True, **kw)),
        adopt_text_dropped_ms=timed(lambda: ops.adopt_step(p, g, m, v, 3, active_b=False, **kw)),
        adopt_ema_ms=timed(lambda: ops.adopt_step(p, g, m, v, 3, active_b=True, ema=e, ema_decay=0.999, **kw)),
        ema_ms=timed(lambda: ops.ema_update(e, p, 0.999)))
    out['GBps'] = dict(sumsq=N * 4 / out['sumsq_ms'] / 1e6, adopt=N * 28 / out['adopt_ms'] / 1e6, adopt_ema=N * 36 / out['adopt_ema_ms'] / 1e6,
                       ema=N * 12 / out['ema_ms'] / 1e6)
    print(json.dumps(out))
    sys.exit(0)

res = {}
for name, var, grid in [('round-robin, nt, fast math, grid 65536', 0, 65536), ('spans', 1, 65536), ('temporal', 2, 65536), ('exact math', 4, 65536),
                        ('round 4: spans, temporal, exact, grid 4096', 7, 4096), ('grid 4096', 0, 4096), ('grid 16384', 0, 16384), ('grid 32768', 0, 32768),
                        ('grid 131072', 0, 131072), ('grid 400000', 0, 400000), ('round-robin, nt, fast math, grid 65536 (again)', 0, 65536)]:
    env = dict(os.environ, E2K_OPTIM_VARIANT=str(var), E2K_OPTIM_GRID=str(grid))
    r = subprocess.run([sys.executable, __file__, '--child'], env=env, capture_output=True, text=True, timeout=300)
    try:
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:       # noqa: BLE001
        res[name] = dict(error=(r.stderr or r.stdout)[-400:])
    print(name, json.dumps(res[name]), flush=True)
if len(sys.argv) > 1:
    json.dump(dict(n=N, variants=res), open(sys.argv[1], 'w'), indent=1)
