"""Probe: does torch.cuda.MemPool / use_mem_pool give a private, address-stable arena on ROCm (needed by the native
launch-plan replay: buffers recorded once must stay reserved for the plan)?"""
import torch
dev = torch.device('cuda')
ok = hasattr(torch.cuda, 'MemPool') and hasattr(torch.cuda, 'use_mem_pool')
print('has MemPool api', ok)
if ok:
    pool = torch.cuda.MemPool()
    with torch.cuda.use_mem_pool(pool):
        a = torch.empty(1 << 20, device=dev)
        pa = a.data_ptr()
        del a
        b = torch.empty(1 << 20, device=dev)
        print('reuse inside pool', b.data_ptr() == pa)
        pb = b.data_ptr()
        del b
    c = torch.empty(1 << 20, device=dev)
    print('outside alloc avoids pool block', c.data_ptr() != pb)
    with torch.cuda.use_mem_pool(pool):
        d = torch.empty(1 << 20, device=dev)
        print('pool block reused on re-entry', d.data_ptr() == pb)
import time
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    x.add_(1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('torch add_ host us/launch', (t1 - t0) / 2000 * 1e6)
