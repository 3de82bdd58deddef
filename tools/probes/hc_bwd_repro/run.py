"""Minimal two-kernel reproducer of the hc_bwd co-residency anomaly: hc_bwd_kernel as it was before
commit c161093 (d(Wp) flushed with LDS float atomics, ds_add_f32; built by build.sh into libhc_r02.so) and today's kernel
(plain LDS stores per token slot), each launched on the default stream while an LDS-DMA NT GEMM of THIS library runs on
another stream with disjoint buffers; dR / dy compared bit for bit with a run alone.  -> gpurun_out/hc_bwd_repro.json"""
import ctypes, json, sys
from pathlib import Path
HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
old = ctypes.CDLL(str(HERE / 'libhc_r02.so'))
old.e2k_hc_bwd.restype = ctypes.c_int
old.e2k_query_hc_bwd_blocks.restype = ctypes.c_int
old.e2k_query_hc_partial_stride.restype = ctypes.c_int
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
res = {}
for M in (928, 960, 1024):
    torch.manual_seed(0)
    D = 512
    X = torch.randn(M, 4, D, device=dev).to(bf16)
    params = [torch.ones(4, device=dev), torch.randn(4, 5, device=dev), torch.randn(D, 5, device=dev) * 0.03,
              torch.tensor(0.01, device=dev), torch.randn(D, device=dev) * 0.03, torch.tensor(0.01, device=dev), torch.zeros(D, device=dev)]
    M1, _, c1 = ops.hc_fwd(X, params)
    y1 = torch.randn(M, D, device=dev).to(bf16)
    _, _, c2 = ops.hc_fwd(M1, params, yprev=y1, coef_prev=c1)
    G = torch.randn(M, 4, D, device=dev).to(bf16); db = torch.randn(M, D, device=dev).to(bf16); y2 = torch.randn(M, D, device=dev).to(bf16)
    grads = [torch.zeros_like(p) for p in params]
    an = torch.randn(2048, 1024, device=dev).to(bf16); wn = torch.randn(2048, 1024, device=dev).to(bf16)
    on = torch.empty(2048, 2048, device=dev, dtype=bf16)
    side = torch.cuda.Stream()
    nb = old.e2k_query_hc_bwd_blocks(M, D)
    partial = torch.zeros(nb * old.e2k_query_hc_partial_stride(D), device=dev)

    def old_bwd():
        dR, dy = torch.empty_like(M1), torch.empty(M, D, device=dev, dtype=bf16)
        st = torch.cuda.current_stream().cuda_stream
        rc = old.e2k_hc_bwd(P(M1), P(y1), P(c1), P(G), P(db), P(y2), P(c2), P(dR), P(dy), *[P(p) for p in params], *[P(g) for g in grads],
                            P(partial), M, D, 1, 1, ctypes.c_void_p(st))
        assert rc == 0, rc
        return dR, dy

    def new_bwd():
        return ops.hc_bwd(G, xin=M1, yprev=y1, coef_prev=c1, dbin=db, ycur=y2, coef=c2, params=params, grads=grads)

    def run(fn, co):
        torch.cuda.synchronize()
        if co:
            with torch.cuda.stream(side):
                for _ in range(8):
                    ops.gemm_nt(an, wn, out=on)
        out = fn()
        torch.cuda.synchronize()
        return [t.clone() for t in out if torch.is_tensor(t)]

    row = {}
    for tag, fn in (('r02_kernel_with_lds_float_atomics', old_bwd), ('current_kernel', new_bwd)):
        ref = run(fn, False)
        alone = sum(any(not torch.equal(a, b) for a, b in zip(run(fn, False), ref)) for _ in range(50))
        co = sum(any(not torch.equal(a, b) for a, b in zip(run(fn, True), ref)) for _ in range(150))
        row[tag] = dict(differing_alone_of_50=alone, differing_next_to_nt_gemm_of_150=co)
    res[f'Mtok={M}'] = row
    print(M, row, flush=True)
(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(res, open(ROOT / 'gpurun_out' / 'hc_bwd_repro.json', 'w'), indent=1)
