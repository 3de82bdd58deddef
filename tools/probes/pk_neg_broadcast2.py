"""Second attempt at a two-kernel reproducer for profiles/r06t_rotary_form_vs_lanes.txt: e2k_qkv_post_fwd (E2K_LIB = the formB variant) in a tight
loop on one stream while another stream runs nothing but LDS-DMA GEMMs of the shapes the test model's text branch launches (128 x 128 kernel,
E2K_GEMM_FLAGS=256) or the 256 x 256 kernel -- every output of every call against the first call's."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / 'e2-tts-pytorch_amd'))
from e2_tts_pytorch_amd import ops                     # noqa: E402

dev, bf16 = 'cuda', torch.bfloat16
torch.manual_seed(0)
B, H, N = 4, 8, 232
I = H * 64
qkvg = torch.randn(B * N, 3 * I + 2 * H, device=dev).to(bf16)
vfirst = torch.randn(B, H, N, 64, device=dev).to(bf16)
cosb, sinb = ops.rotary_table(N, dev)
side = torch.cuda.Stream()
shapes = [(928, 1552, 512), (928, 512, 512), (928, 776, 256), (928, 2048, 256), (8448, 1024, 1024)]
gem = [(torch.randn(m, k, device=dev).to(bf16), torch.randn(n, k, device=dev).to(bf16)) for m, n, k in shapes]
INNER = int(os.environ.get('INNER', '24'))


def post():
    st = ops.qkv_post_fwd(qkvg, B, H, N, cosb, sinb, vfirst, need_v=False)
    return st.Q, st.K, st.VT


ref = [t.clone() for t in post()]
torch.cuda.synchronize()
for mode in ('alone', 'next to LDS-DMA GEMMs'):
    bad, nel, examples = 0, [0, 0, 0], []
    for it in range(int(os.environ.get('ITERS', '200'))):
        if mode != 'alone':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(4):
                    for a, w in gem:
                        ops.gemm_nt(a, w)
        outs = [post() for _ in range(INNER)]
        for got in outs:
            d = [int((g != r).sum()) for g, r in zip(got, ref)]
            bad += any(d)
            for i in range(3):
                nel[i] += d[i]
            if any(d) and len(examples) < 3:
                i = max(range(3), key=lambda j: d[j])
                idx = (got[i] != ref[i]).nonzero()
                examples.append((('Q', 'K', 'VT')[i], d, idx[:3].tolist(), [float(got[i][tuple(j)]) for j in idx[:3]], [float(ref[i][tuple(j)]) for j in idx[:3]]))
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(f'{mode}: {bad} of {INNER * int(os.environ.get("ITERS", "200"))} calls differ from the first; differing elements of Q / K / V^T in total {nel}; examples (tensor, counts, indices, got, first call) {examples}', flush=True)
