"""gpurun_out/nt_traffic_{FETCH,WRITE}_SIZE.csv (tools/gpu/traffic.sh) -> profiles/r06_nt_traffic.json

One main-kernel dispatch (+ its fix-up dispatch, if any) per entry of tools/nt_shapes_cfg3.json, in file order; the
per-launch numbers are weighted by the call count of each shape in a cfg3 training step."""
import csv, json, re
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
shapes = json.load(open(ROOT / 'tools' / 'nt_shapes_cfg3.json'))


def per_shape(fn):
    d = []
    for r in csv.DictReader(open(fn)):
        if re.search(r'gemm_nt_glds_kernel|gemm_nt_fixup_kernel|gemm_nt_kernel|gemm_nt_256_kernel|gemm_nt_256_fixup_kernel', r['Kernel_Name']):
            d.append((int(r['Dispatch_Id']), 'fixup' in r['Kernel_Name'], float(r['Counter_Value'])))
    d.sort()
    g = []
    for _, fix, v in d:
        if fix:
            g[-1] += v
        else:
            g.append(v)
    return g


gf = per_shape(ROOT / 'gpurun_out' / 'nt_traffic_FETCH_SIZE.csv')
gw = per_shape(ROOT / 'gpurun_out' / 'nt_traffic_WRITE_SIZE.csv')
assert len(gf) == len(shapes) == len(gw), (len(gf), len(gw), len(shapes))
n = sum(s['count'] for s in shapes)
rows, fetch, write, alg = [], 0.0, 0.0, 0.0
for s, f, w in zip(shapes, gf, gw):
    K = s['K1'] + s['K2']
    a = (s['M'] * K + s['N'] * K) * 2 + s['M'] * s['N'] * (4 if s['out_f32'] else 2) + (s['M'] * s['N'] * 2 if s['resid'] else 0)
    fb, wb = f * 1024 * 2, w * 1024        # KB -> bytes; gfx950 FETCH_SIZE tallies 128-B requests at 64 B (x2)
    rows.append(dict(s, fetch_bytes=fb, write_bytes=wb, algorithmic_bytes=a))
    fetch += fb * s['count']; write += wb * s['count']; alg += a * s['count']
res = dict(kernel='gemm_nt_256_kernel / gemm_nt_glds_kernel (+fix-ups): the NT launch mix of a cfg3 step', launches_per_step=n, hbm_fetch_bytes_per_launch=fetch / n,
           hbm_write_bytes_per_launch=write / n, traffic_bytes_per_launch=(fetch + write) / n,
           algorithmic_bytes_per_launch=alg / n,
           method='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over tools/nt_traffic_probe.py: one launch '
                  'per distinct NT shape of a cfg3 step (tools/nt_shapes_cfg3.json), weighted by its call count; FETCH_SIZE (KB) '
                  'doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE (KB) uncalibrated',
           shapes=rows)
json.dump(res, open(ROOT / 'profiles' / 'r06_nt_traffic.json', 'w'), indent=1)
print({k: v for k, v in res.items() if k not in ('shapes', 'method')})
