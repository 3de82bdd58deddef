"""How much of the NT GEMM launch mix's L2-miss traffic (profiles/r03_nt_traffic.json: rocprofv3 --pmc FETCH_SIZE per shape) is
forced by the machine?  Each XCD has its own 4-MB L2 and runs a contiguous run of the tile list (xcd_remap + tile_coords,
csrc/gemm.hip): its resident set of a x b tiles (32 tiles of 256 x 256 at one workgroup per CU, 64 of 128 x 128 at two; 8
M-tiles per group) has to bring a + b operand panels of tile x K in from outside the L2, and a panel set (2-6 MB) does not
survive until the next resident set.  Lower bound per shape = tiles / (a b) resident sets x (a + b) panels (not below the
algorithmic A + B), + the residual read.  Writes profiles/r03_nt_traffic_bound.json.  (FETCH_SIZE counts L2 misses whether HBM or the
256-MB Infinity Cache serves them: the HBM bytes proper are not observable from the TCC counters.)"""
import json
import math
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
d = json.load(open(ROOT / 'profiles' / 'r03_nt_traffic.json'))
rows, tm_, tb_, ta_ = [], 0.0, 0.0, 0.0
for s in d['shapes']:
    M, N, K = s['M'], s['N'], s['K1'] + s['K2']
    big = math.ceil(M / 256) * math.ceil(N / 256) >= 224 and K >= 256           # the host's kernel choice (gemm.hip)
    T = 256 if big else 128
    tm, tn = math.ceil(M / T), math.ceil(N / T)
    conc = 32 if big else 64
    a = min(8, tm)
    b = max(1, min(tn, conc // a))
    if a * b < conc and b == tn:
        a *= conc // (a * b)
    resid = M * N * 2 if s['resid'] else 0
    alg = (M * K + N * K) * 2 + resid
    bound = max((M * K + N * K) * 2, (a + b) * T * K * 2 * tm * tn / (a * b)) + resid
    rows.append(dict(M=M, N=N, K=K, tile=T, count=s['count'], resident_set=f'{a} x {b}', fetch_MB=round(s['fetch_bytes'] / 1e6, 1),
                     bound_MB=round(bound / 1e6, 1), algorithmic_read_MB=round(alg / 1e6, 1)))
    tm_ += s['fetch_bytes'] * s['count']
    tb_ += bound * s['count']
    ta_ += alg * s['count']
out = dict(measured_fetch_over_algorithmic_reads=round(tm_ / ta_, 3), l2_capacity_bound_over_algorithmic_reads=round(tb_ / ta_, 3),
           measured_fetch_over_bound=round(tm_ / tb_, 3), shapes=rows,
           note='weighted by launches per step; reads only (the write side of roofline.traffic is the C tiles, written once)')
(ROOT / 'profiles' / 'r03_nt_traffic_bound.json').write_text(json.dumps(out, indent=1) + '\n')
print({k: v for k, v in out.items() if k != 'shapes'})
