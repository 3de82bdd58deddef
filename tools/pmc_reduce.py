"""gpurun_out/pmc{,2,3}_<what>_counters.csv (tools/gpu/pmc.sh <what>) -> profiles/<round>_<what>_pmc.json: per kernel, the average per
dispatch of every counter (whole chip), plus two derived figures for the attention kernels (vector-pipe busy fraction, vector
instructions per wave).

    python tools/pmc_reduce.py attn r04
"""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
what, rnd = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for k in ('pmc', 'pmc2', 'pmc3'):
    f = ROOT / 'gpurun_out' / f'{k}_{what}_counters.csv'
    if not f.exists():
        continue
    per = defaultdict(float)          # (dispatch, kernel, counter) -> value summed over the rows rocprofv3 splits a dispatch into
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name'])
        name = re.sub(r'\(.*$', '', name)
        per[(r['Dispatch_Id'], name, r['Counter_Name'])] += float(r['Counter_Value'])
    for (_, name, c), v in per.items():
        acc[name][c].append(v)
out = {'note': f'rocprofv3 --pmc passes (tools/gpu/pmc.sh {what}) over tools/gemm_probe.py {what}; averages per launch, whole chip; '
               'SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are in quad-cycles summed over waves', 'kernels': {}}
for name, cs in sorted(acc.items()):
    if name.startswith('at::') or 'elementwise_kernel' in name:
        continue          # torch's own kernels of the probe's set-up
    d = {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}
    if d.get('SQ_WAVES') and d.get('SQ_INSTS_VALU'):
        d['derived_valu_instr_per_wave'] = round(d['SQ_INSTS_VALU'] / d['SQ_WAVES'])
    if d.get('SQ_ACTIVE_INST_VALU') and d.get('SQ_BUSY_CYCLES'):
        # SQ_BUSY_CYCLES counts per SE-level SQ; the round-2 summary normalised by 9 x SQ_BUSY_CYCLES (kept for comparability)
        d['derived_valu_pipe_busy_frac'] = round(d['SQ_ACTIVE_INST_VALU'] / (9.0 * d['SQ_BUSY_CYCLES']), 3)
    out['kernels'][name] = d
dst = ROOT / 'profiles' / f'{rnd}_{what}_pmc.json'
json.dump(out, open(dst, 'w'), indent=1)
print(dst, list(out['kernels']))
