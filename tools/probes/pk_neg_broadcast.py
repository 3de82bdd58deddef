"""Does e2k_qkv_post_fwd reproduce its own output next to a GEMM on another stream?  (MI355X; E2K_LIB selects the library: today's, or the
variant whose rotary pair is written  b = fma(x1, c, x0 s)  and compiles to `v_pk_mul_f32 .. op_sel_hi:[0,1] neg_hi:[1,0]`.)"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / 'e2-tts-pytorch_amd'))
from e2_tts_pytorch_amd import ops                     # noqa: E402

dev = 'cuda'
bf16 = torch.bfloat16
torch.manual_seed(0)
B, H, N = 4, 8, 232
I = H * 64
cols = 3 * I + 2 * H
qkvg = torch.randn(B * N, cols, device=dev).to(bf16)
vfirst = torch.randn(B, H, N, 64, device=dev).to(bf16)
cosb, sinb = ops.rotary_table(N, dev)
a = torch.randn(8448, 1024, device=dev).to(bf16)
w = torch.randn(4096, 1024, device=dev).to(bf16)
side = torch.cuda.Stream()
Dt = 256
xs = torch.randn(B * N, 4, Dt, device=dev).to(bf16)
hcp = [torch.randn(4, device=dev), torch.randn(4, 5, device=dev), torch.randn(Dt, 5, device=dev) * Dt ** -0.5, torch.full((), 0.3, device=dev),
       torch.randn(Dt, device=dev) * Dt ** -0.5, torch.full((), 0.3, device=dev), torch.randn(Dt, device=dev) * 0.2]
gam = torch.ones(1, Dt, device=dev)
H2 = 4
qkvg2 = torch.randn(B * N, 3 * H2 * 64 + 2 * H2, device=dev).to(bf16)
vfirst2 = torch.randn(B, H2, N, 64, device=dev).to(bf16)
kmask = torch.ones(B, (N + 63) // 64 * 64, dtype=torch.uint8, device=dev)
kmask[:, N:] = 0
a2 = torch.randn(B * N, 256, device=dev).to(bf16)
w2 = torch.randn(776, 256, device=dev).to(bf16)


CFG = os.environ.get('CFG', 'later_layer')       # later_layer: value residual given; first_layer: none; nograd: value residual, V not written


def post():
    st = ops.qkv_post_fwd(qkvg if CFG != 'first_layer' else qkvg[:, :3 * I + H], B, H, N, cosb, sinb, None if CFG == 'first_layer' else vfirst,
                          need_v=CFG != 'nograd')
    z = torch.zeros(1, device=dev)
    return (st.Q.clone(), st.K.clone(), st.V.clone() if st.V is not None else z, st.VT.clone(), st.gate.clone(),
            st.mix.clone() if st.mix is not None else z)


ref = post()
torch.cuda.synchronize()
for mode in ('alone', 'next to a GEMM on another stream', 'next to post + attention + a small GEMM + a width connection on another stream'):
    bad = [0, 0, 0, 0, 0, 0]
    nbad_el = 0
    for it in range(int(os.environ.get('ITERS', '300'))):
        if mode != 'alone':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if mode.startswith('next to a GEMM'):
                    for _ in range(3):
                        ops.gemm_nt(a, w)
                else:               # what the TEXT lane runs next to it in the backbone: the same kernels on the text branch's buffers
                    for _ in range(2):
                        st2 = ops.qkv_post_fwd(qkvg2, B, H2, N, cosb, sinb, vfirst2)
                        ops.attn_fwd(st2, kmask)
                        ops.gemm_nt(a2, w2)
                        ops.hc_fwd(xs, hcp, norm=(gam, 0., B * N), want_bin=False)
        got = post()
        for i, (g, r) in enumerate(zip(got, ref)):
            if not torch.equal(g, r):
                bad[i] += 1
                if i < 2:
                    nbad_el += int((g != r).sum())
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(f'{mode}: {CFG}: calls whose Q / K / V / V^T / gate / mix differ from the first call: {bad}; differing Q / K elements in total {nbad_el}', flush=True)
