"""Kernels must give the same bits whether or not another kernel shares the chip (launch lanes put the text branches and
the weight-gradient GEMMs on side streams).  hc_bwd_kernel once did not: with LDS float atomics (ds_add_f32) in its
gradient flush, a few tokens per launch came out 1-4 bf16 ulp off whenever an LDS-DMA GEMM ran next to it -- disjoint
buffers, unchanged inputs (tools/probes/hc_concurrent.py, profiles/r02_launch_lanes.json).  GPU only: the host model
runs one kernel at a time."""
import pytest
import torch


def install_lib(path, host_pointers):
    from emu.install import install
    install(path, host_pointers)


bf16 = torch.bfloat16


@pytest.mark.gpu
def test_results_do_not_depend_on_a_concurrent_gemm():
    from e2_tts_pytorch_amd import ops, _lib
    install_lib(None, host_pointers=False)
    dev = 'cuda'
    torch.manual_seed(0)
    M, D = 960, 512
    X = torch.randn(M, 4, D, device=dev).to(bf16)
    params = [torch.ones(4, device=dev), torch.randn(4, 5, device=dev), torch.randn(D, 5, device=dev) * 0.03,
              torch.tensor(0.01, device=dev), torch.randn(D, device=dev) * 0.03, torch.tensor(0.01, device=dev), torch.zeros(D, device=dev)]
    M1, _, c1 = ops.hc_fwd(X, params)
    y1 = torch.randn(M, D, device=dev).to(bf16)
    _, _, c2 = ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1)
    G = torch.randn(M, 4, D, device=dev).to(bf16)
    db = torch.randn(M, D, device=dev).to(bf16)
    y2 = torch.randn(M, D, device=dev).to(bf16)
    grads = [torch.zeros_like(p) for p in params]
    xx = torch.randn(8, 120, D, device=dev).to(bf16)
    cw, cb = torch.randn(D, 31, device=dev) * 0.1, torch.zeros(D, device=dev)
    pre, _ = ops.dwconv_fwd(xx, None, cw, cb)
    dwg, dbg = torch.zeros(D, 31, device=dev), torch.zeros(D, device=dev)
    an = torch.randn(2048, 1024, device=dev).to(bf16)
    wn = torch.randn(2048, 1024, device=dev).to(bf16)
    on = torch.empty(2048, 2048, device=dev, dtype=bf16)
    at = torch.randn(1024, 1552, device=dev).to(bf16)
    bt = torch.randn(1024, 512, device=dev).to(bf16)
    ot = torch.zeros(1552, 512, device=dev)
    side = torch.cuda.Stream()

    victims = {
        'hc_bwd': lambda: ops.hc_bwd(G, xin=M1, yprev=y1, coef_prev=c1, dbin=db, ycur=y2, coef=c2, params=params, grads=grads),
        'hc_fwd': lambda: ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1),
        'dwconv_bwd': lambda: (ops.dwconv_bwd(xx, pre, xx, None, cw, dwg, dbg),),
    }
    corunners = {
        'NT GEMM (global_load_lds)': lambda: ops.gemm_nt(an, wn, out=on),
        'TN GEMM (global_load_lds + ds_read_b64_tr_b16)': lambda: ops.gemm_tn(at, bt, ot),
    }

    def run(fn, co):
        torch.cuda.synchronize()
        if co is not None:
            with torch.cuda.stream(side):
                for _ in range(8):
                    co()
        out = fn()
        torch.cuda.synchronize()
        return [t.clone() for t in out if torch.is_tensor(t)]

    bad = {}
    for vn, fn in victims.items():
        ref = run(fn, None)
        for cn, co in corunners.items():
            n = sum(any(not torch.equal(a, b) for a, b in zip(run(fn, co), ref)) for _ in range(40))
            if n:
                bad[vn, cn] = n
    assert not bad, bad


@pytest.mark.gpu
def test_data_parallel_hook_waits_for_the_launch_lanes():
    """ddp.DataParallel on one rank (RCCL, world size 1): the slab all-reduce runs on its own stream and must not start
    before the TEXT and WGRAD lanes have finished the layer -- same gradients as the unwrapped module, replayed plans"""
    import copy
    import os
    import random
    import torch.distributed as dist
    from e2_tts_pytorch_amd import Transformer, _lib
    from e2_tts_pytorch_amd.ddp import DataParallel
    from test_backbone import randomize, rel2
    install_lib(None, host_pointers=False)
    dev = 'cuda'
    random.seed(0)
    torch.manual_seed(0)
    dim, depth, B, T = 512, 4, 4, 200
    plain = Transformer(dim=dim, depth=depth, heads=dim // 64, dropout=0., max_seq_len=T)
    randomize(plain)
    plain = plain.to(dev)
    wrapped_mod = copy.deepcopy(plain)
    R = torch.randn(B, T, dim, device=dev)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{29600 + os.getpid() % 300}', rank=0, world_size=1)
    try:
        net = DataParallel(wrapped_mod, grad_dtype=torch.float32)

        def step(m, seed):
            m.zero_grad(set_to_none=True)
            g = torch.Generator().manual_seed(seed)
            x = torch.randn(B, T, dim, generator=g).to(dev)
            t = torch.rand(B, generator=g).to(dev)
            txt = torch.randn(B, T, dim // 2, generator=g).to(dev)
            (m(x, times=t, text_embed=txt) * R).sum().backward()
            torch.cuda.synchronize()
            return {n: p.grad.clone() for n, p in (m.module if isinstance(m, DataParallel) else m).named_parameters()}

        for seed in (1, 2, 3, 4):              # first sighting, recording, two replays
            g0, g1 = step(plain, seed), step(net, seed)
            assert net._sync.lanes and net._sync.calls > 0
            bad = [(n, rel2(g1[n], g0[n])) for n in g0 if float(g0[n].norm()) > 1e-6 and rel2(g1[n], g0[n]) > 2e-3]
            assert not bad, (seed, bad[:5])
    finally:
        if created:
            dist.destroy_process_group()
