"""Probe (GPU): the root cause of the round-3 GPU test abort (GPUTEST_r03: SIGABRT inside a recorded backward pass).

`torch.cuda.MemPool.__del__` -> emptyCache(pool) -> the allocator asserts `captures_underway.empty()`
(c10/hip/HIPCachingAllocator.cpp).  `torch.cuda.use_mem_pool` IS such a "capture under way", so a MemPool that is
destroyed while any other pool context is open takes the process down (the assert throws inside a destructor ->
std::terminate -> SIGABRT).  A launch plan of this package records inside `use_mem_pool`; a plan of a dead module is cyclic
garbage, i.e. its pool dies whenever Python's cycle collector happens to run.

Three child processes, exit code and the last stderr line of each:
  torch_only      the torch behaviour by itself, no code of this package
  product_r03     this package with the round-3 behaviour restored (pool dies where its plan dies), collector forced to run
                  inside the next module's recording  -> expected: SIGABRT (-6 / 134)
  product_now     the same with today's `_PlanPool` (destruction parked until no pool context is open) -> expected: 0

  python tools/probes/mempool_dtor_abort.py [out.json]
"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent

TORCH_ONLY = r'''
import torch
p1 = torch.cuda.MemPool()
with torch.cuda.use_mem_pool(p1):
    a = torch.empty(1 << 20, device='cuda')
del a
p2 = torch.cuda.MemPool()
with torch.cuda.use_mem_pool(p2):
    b = torch.empty(1 << 20, device='cuda')
    del p1          # MemPool destructor inside another pool's context
torch.cuda.synchronize()
print('survived')
'''

PRODUCT = r'''
import gc, random, sys
sys.path[:0] = [%(pkg)r, %(root)r]
import torch
from e2_tts_pytorch_amd import Transformer, ops, backbone
if %(old)r:
    del backbone._PlanPool.__del__        # round 3: nothing keeps the MemPool alive past its plan
random.seed(0); torch.manual_seed(0)

def steps(m, n):
    for i in range(n):
        x = torch.randn(2, 24, 256, device='cuda', requires_grad=True)
        m(x, times=torch.rand(2, device='cuda'), text_embed=torch.randn(2, 24, 128, device='cuda')).sum().backward()

gc.disable()
a = Transformer(dim=256, depth=2, heads=2, dropout=0., max_seq_len=64).cuda()
steps(a, 3)                               # first sighting, recording, replay: `a` owns a plan and its pool
assert any(not isinstance(v, str) for v in a._plans.values())
a.__dict__['_me'] = a                     # cyclic garbage: only the collector can free it
del a
b = Transformer(dim=256, depth=2, heads=2, dropout=0., max_seq_len=64).cuda()
orig = ops.begin_recording
def hooked(meta=None):
    orig(meta)
    gc.collect()                          # the collector strikes inside b's recording = inside a use_mem_pool context
ops.begin_recording = hooked
steps(b, 3)
torch.cuda.synchronize()
print('survived')
'''


def run(code):
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    err = [ln for ln in r.stderr.strip().splitlines() if ln.strip()]
    key = [ln for ln in err if 'INTERNAL ASSERT' in ln or 'terminate called' in ln or 'captures_underway' in ln]
    return dict(rc=r.returncode, stdout=r.stdout.strip()[-200:], stderr_key_lines=key[:4], stderr_tail=err[-3:])


def main():
    args = dict(pkg=str(ROOT / 'e2-tts-pytorch_amd'), root=str(ROOT))
    out = dict(torch_only=run(TORCH_ONLY),
               product_r03=run(PRODUCT % dict(args, old=True)),
               product_now=run(PRODUCT % dict(args, old=False)))
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
