"""out-of-bounds check of the weight-gradient GEMM: operands, output, workspace and column sums inside canary-filled buffers"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops, _lib
bf16, f32 = torch.bfloat16, torch.float32
dev = 'cuda'
PAD = 1 << 16
def guarded(n, dtype, fill):
    buf = torch.full((n + 2 * PAD,), fill, dtype=dtype, device=dev)
    return buf, buf[PAD:PAD + n]
def check(buf, n, fill, name):
    lo, hi = buf[:PAD], buf[PAD + n:]
    bad = int((lo != fill).sum() + (hi != fill).sum())
    if bad:
        print('   OOB WRITE around', name, bad, 'elements', flush=True)
    return bad
L = _lib.get()
for (M, N, K, csf) in [(1024, 1552, 512, 1536), (1024, 1552, 512, None), (1024, 2730, 512, 0), (1024, 512, 1365, 0), (1024, 776, 256, 768), (8448, 3104, 1024, 3072), (8448, 1552, 512, 1536), (8448, 512, 512, None), (8448, 1024, 512, None), (8448, 8192, 1024, 0), (33792, 1024, 512, None), (928, 1552, 512, 1536), (928, 2730, 512, 0), (928, 512, 1365, 0), (928, 512, 512, None), (464, 776, 256, 768), (928, 776, 256, 768)]:
    lda, ldb = (N + 7) // 8 * 8, (K + 7) // 8 * 8
    abuf, a = guarded(M * lda, bf16, 7.0); bbuf, b = guarded(M * ldb, bf16, 7.0)
    a.normal_(); b.normal_()
    A, B = a.view(M, lda)[:, :N], b.view(M, ldb)[:, :K]
    obuf, o = guarded(N * K, f32, 7.0); o.zero_()
    cbuf, c = guarded(N, f32, 7.0); c.zero_()
    ns = L.e2k_query_gemm_tn_splits_mode(M, N, K, 0, 1)
    wbuf, w = guarded(max(ns, 1) * N * K, f32, 7.0)
    rc = L.e2k_gemm_tn_bf16(A.data_ptr(), lda, B.data_ptr(), ldb, o.data_ptr(), K, M, N, K, 0, 1, w.data_ptr() if ns > 1 else None,
                            c.data_ptr() if csf is not None else None, csf or 0, None)
    torch.cuda.synchronize()
    ref = A.float().T @ B.float()
    err = float((o.view(N, K) - ref).abs().max() / ref.abs().max())
    bad = check(abuf, M * lda, 7.0, 'A') + check(bbuf, M * ldb, 7.0, 'B') + check(obuf, N * K, 7.0, 'C') + check(cbuf, N, 7.0, 'colsum') + check(wbuf, max(ns, 1) * N * K, 7.0, 'ws')
    print((M, N, K, csf), 'rc', rc, 'splits', ns, 'err', err, 'oob', bad, flush=True)
