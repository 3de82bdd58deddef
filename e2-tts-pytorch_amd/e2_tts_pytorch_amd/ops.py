"""Thin torch-tensor front end of the e2k C ABI: argument checking, raw pointers, current HIP stream.

PyTorch is plumbing here (device memory + stream); all arithmetic happens in csrc/*.hip.
"""
from __future__ import annotations

import torch

from . import _lib

bf16 = torch.bfloat16
f32 = torch.float32


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda or _lib.host_pointers_ok()):
            raise _lib.E2KError('e2_tts_pytorch_amd kernels need tensors on a HIP device (no CPU path)')


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None


def _rows(t):
    """(rows, row_stride) of a 2-D view whose last dim is contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.shape[0], t.stride(0)


# ------------------------------------------------------------------------------------------------ GEMM

def gemm_nt(a, b, *, a2=None, out=None, out_dtype=bf16, accumulate=False, bias=None, colscale=None,
            rows_per_batch=0, rowmask=None, resid=None):
    """out[M,N] = ((([a|a2] @ b.T) + bias) * colscale[row // rows_per_batch]) * rowmask[:,None] + resid"""
    _chk(a, b, a2, out, bias, colscale, rowmask, resid)
    assert a.dtype == bf16 and b.dtype == bf16
    M, lda = _rows(a)
    N, ldb = _rows(b)
    K1 = a.shape[1]
    K2, lda2 = 0, 0
    if a2 is not None:
        assert a2.dtype == bf16 and a2.shape[0] == M
        _, lda2 = _rows(a2)
        K2 = a2.shape[1]
    assert b.shape[1] == K1 + K2, (a.shape, None if a2 is None else a2.shape, b.shape)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype in (bf16, f32)
    if bias is not None:
        assert bias.dtype == f32 and bias.numel() == N and bias.is_contiguous()
    if colscale is not None:
        assert colscale.dtype == f32 and colscale.is_contiguous() and colscale.shape[-1] == N and rows_per_batch > 0
    if rowmask is not None:
        assert rowmask.dtype in (torch.uint8, torch.bool) and rowmask.numel() == M and rowmask.is_contiguous()
    ldr = 0
    if resid is not None:
        assert resid.dtype == bf16 and resid.shape == (M, N) and resid.stride(1) == 1
        ldr = resid.stride(0)
    _lib.get().e2k_gemm_nt_bf16(_p(a), lda, K1, _p(a2), lda2, K2, _p(b), ldb, _p(out), out.stride(0),
                                int(out.dtype == f32), int(accumulate), M, N, _p(bias), _p(colscale),
                                int(rows_per_batch), _p(rowmask), _p(resid), ldr, _stream(a))
    return out


def gemm_tn(a, b, out, *, splits=0, use_tr=True):
    """out[N,K] += a[M,N].T @ b[M,K]   (fp32 out, bf16 a/b)"""
    _chk(a, b, out)
    assert a.dtype == bf16 and b.dtype == bf16 and out.dtype == f32
    M, lda = _rows(a)
    M2, ldb = _rows(b)
    assert M == M2
    N, K = a.shape[1], b.shape[1]
    assert out.shape == (N, K) and out.stride(1) == 1
    _lib.get().e2k_gemm_tn_bf16(_p(a), lda, _p(b), ldb, _p(out), out.stride(0), M, N, K, int(splits),
                                int(use_tr), _stream(a))
    return out
