"""the optimizer-side kernels alone at cfg3 size (716 M fp32 elements per buffer), HIP events.   python tools/probes/optim_ab.py [out.json]"""
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

# (round 5 ran this once per variant through E2K_OPTIM_VARIANT / E2K_OPTIM_GRID switches -- block mapping, non-temporal accesses, fast
#  sqrt / rcp, grid cap: profiles/r05k_optim_kernels_ab.json, r05m_optim_kernels_ab.json; the switches were removed with the losers)
r = subprocess.run([sys.executable, __file__, '--child'], capture_output=True, text=True, timeout=300)
res = json.loads(r.stdout.strip().splitlines()[-1])
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(dict(n=N, kernels=res), open(sys.argv[1], 'w'), indent=1)
