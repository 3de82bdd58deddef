"""round 6: where the host time of a replayed step goes -- the launch-floor probe of bench.py (a tiny model with the launch count of
cfg3, negligible kernel time) in four modes: launch lanes on / off (off = no event record / wait operations at all) x eager replay /
HIP-graph replay.  Run on the GPU box: python tools/probes/host_floor_modes.py [depth]"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, json, torch
sys.path[:0] = [%(root)r, %(root)r + '/e2-tts-pytorch_amd']
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
r = bench.host_launch_floor(%(depth)d, dev, 0.1, graphs=%(graphs)r)
print('RESULT ' + json.dumps(r))
'''

if __name__ == '__main__':
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    out = {}
    for lanes in ('3', '0'):
        for graphs in (False, True):
            env = dict(os.environ, E2K_LANES=lanes, GPU_MAX_HW_QUEUES='8')
            p = subprocess.run([sys.executable, '-c', CHILD % dict(root=str(ROOT), depth=depth, graphs=graphs)], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')]
            out[f'lanes={lanes} graphs={graphs}'] = json.loads(line[0][7:]) if line else p.stderr[-400:]
            print(f'lanes={lanes} graphs={graphs}', out[f'lanes={lanes} graphs={graphs}'], flush=True)
    d = ROOT / 'gpurun_out'
    if d.is_dir():
        json.dump(out, open(d / 'r06e_host_floor_modes.json', 'w'), indent=1)
