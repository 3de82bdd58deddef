"""hardware check of ops.attn_share_dropmask: per-tensor mismatch counts between the re-hash and the shared-mask paths"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
bf16 = torch.bfloat16
dev = 'cuda'
for N in (70, 150):
    torch.manual_seed(0)
    B, H = 2, 3
    I = H * 64
    qkvg = torch.zeros(B * N, (3 * I + H + 7) // 8 * 8, dtype=bf16, device=dev)[:, :3 * I + H]       # padded row stride
    qkvg.copy_(torch.randn(B * N, 3 * I + H).to(bf16))
    cosb, sinb = ops.rotary_table(N, dev)
    kmask = torch.zeros(B, (N + 63) // 64 * 64, dtype=torch.uint8)
    kmask[0, :N] = 1; kmask[1, :N - 9] = 1
    dOg = torch.randn(B * N, I).to(bf16).to(dev)
    res = []
    for share in (False, True):
        ops.attn_share_dropmask = share
        st = ops.qkv_post_fwd(qkvg, B, H, N, cosb, sinb, None)
        Og = ops.attn_fwd(st, kmask.to(dev), 0.25, 1234, 3).clone()
        dQ, dK, dV, dg = ops.attn_bwd(st, dOg, kmask.to(dev), 0.25, 1234, 3)
        res.append(dict(Og=Og, dQ=dQ.clone(), dK=dK.clone(), dV=dV.clone(), dgate=dg.clone()))
        if share:
            w = st.dropbits.view(B * H, -1)
            print('N', N, 'dropbits words', w.numel(), 'nonzero', int((w != 0).sum()))
    ops.attn_share_dropmask = True
    for k in res[0]:
        a, b = res[0][k].float(), res[1][k].float()
        print(' ', k, 'mismatches', int((a != b).sum()), 'of', a.numel(), 'max abs diff', float((a - b).abs().max()), 'max |ref|', float(a.abs().max()))
