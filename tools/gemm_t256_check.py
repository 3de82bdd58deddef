"""First hardware run of the 256 x 256 8-phase NT kernel (flag E2K_GEMM_T256): correctness, race screen, A/B timing.

The kernel was written at the end of round 1 without GPU time left; on the host model it passes with LDS-DMA copies
landing both as early and as late as its counted waits allow (tests/test_kernels_gemm.py::test_gemm_nt_256_tile), which
says nothing about real timing.  This script is the hardware half of that check:

  1. value check against torch.matmul (fp32 reference on the bf16 operands) and against the default 128 x 128 kernel,
     small ragged shapes first (one workgroup, short K) so that a hang or a wrong tile shows up before the big launches;
  2. race screen: the same launch repeated `--reps` times must give bit-identical output every time (a read that beats
     its LDS-DMA copy shows up as an intermittent difference, cdna_hip_programming.md "two-lane discipline");
  3. timing A/B per cfg3 shape: default kernel vs T256, interleaved, HIP events.

    python tools/gemm_t256_check.py [--reps 50] [--iters 20] [--quick]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'e2-tts-pytorch_amd'))

import torch  # noqa: E402

from e2_tts_pytorch_amd import ops  # noqa: E402

T256, NO_T256 = 128, 256
bf16 = torch.bfloat16


def run(a, b, a2, flags, **kw):
    old = ops.gemm_flags
    ops.gemm_flags = flags
    try:
        return ops.gemm_nt(a, b, a2=a2, **kw)
    finally:
        ops.gemm_flags = old


def rel(x, y):
    return ((x.float() - y.float()).abs().max() / y.float().abs().max().clamp_min(1e-20)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--quick', action='store_true', help='value check + race screen on the small shapes only')
    args = ap.parse_args()
    dev = torch.device('cuda')
    torch.manual_seed(0)
    ok = True

    small = [(256, 256, 64, 0), (300, 300, 128, 0), (520, 260, 192, 64), (2304, 256, 512, 0), (700, 520, 320, 0),
             (1280, 512, 256, 256)]
    cfg3 = [(33792, 1024, 1024, 512), (8448, 8192, 1024, 0), (33792, 512, 1024, 512), (8448, 4096, 1024, 0),
            (33792, 1024, 1024, 1024), (8448, 3104, 1024, 0), (33792, 1024, 1024, 0), (8448, 4096, 512, 0),
            (8448, 2048, 512, 0), (8448, 1024, 4096, 0), (8448, 1024, 8192, 0)]
    shapes = small if args.quick else small + cfg3
    rows = []
    for (M, N, K1, K2) in shapes:
        a = torch.randn(M, K1, device=dev).to(bf16)
        a2 = torch.randn(M, K2, device=dev).to(bf16) if K2 else None
        b = torch.randn(N, K1 + K2, device=dev).to(bf16)
        bias = torch.randn(N, device=dev)
        A = torch.cat([a, a2], 1) if K2 else a
        ref = A.float() @ b.float().T + bias
        base = run(a, b, a2, NO_T256, bias=bias)
        out = run(a, b, a2, T256, bias=bias)
        torch.cuda.synchronize()
        e_ref, e_base = rel(out, ref), rel(out, base)
        same = True
        for _ in range(args.reps):
            again = run(a, b, a2, T256, bias=bias)
            if not torch.equal(again, out):
                same = False
                break
        good = e_ref < 6e-3 and same
        ok &= good
        row = dict(M=M, N=N, K1=K1, K2=K2, err_vs_fp32=e_ref, err_vs_default=e_base, repeatable=same, ok=good)
        if not args.quick and (M, N, K1, K2) in cfg3:
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(2)]
            t = [0.0, 0.0]
            for f in (NO_T256, T256):                     # warm
                run(a, b, a2, f, bias=bias)
            for _ in range(args.iters):
                for k, f in enumerate((NO_T256, T256)):
                    ev[k][0].record()
                    run(a, b, a2, f, bias=bias)
                    ev[k][1].record()
                torch.cuda.synchronize()
                for k in range(2):
                    t[k] += ev[k][0].elapsed_time(ev[k][1])
            fl = 2.0 * M * N * (K1 + K2)
            row.update(ms_default=t[0] / args.iters, ms_t256=t[1] / args.iters,
                       tf_default=fl / (t[0] / args.iters * 1e-3) / 1e12, tf_t256=fl / (t[1] / args.iters * 1e-3) / 1e12)
        rows.append(row)
        print(json.dumps(row), flush=True)
    print('T256 CHECK', 'PASSED' if ok else 'FAILED')
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'gemm_t256_check.json'), 'w'), indent=1)
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
