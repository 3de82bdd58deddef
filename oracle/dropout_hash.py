"""TEST INFRASTRUCTURE (oracle side): bit-exact numpy restatement of the counter-hash dropout masks the HIP kernels
generate (e2k_device.h: fmix32 / rand_base / rand_at; attn.hip: drop_scale; elementwise.hip: keep_scale).

The reference draws dropout masks from torch's Philox stream (nn.Dropout inside x_transformers, e2_tts.py:540,641,646);
device RNG streams cannot be matched across implementations, so for parity runs the oracle is fed the masks the
kernels will use.  Integer arithmetic only -> bit-exact.
"""
import numpy as np
import torch

_M = np.uint32


def fmix32(h):
    h = h.astype(np.uint32)
    h ^= h >> _M(16)
    h = (h * _M(0x85ebca6b)).astype(np.uint32)
    h ^= h >> _M(13)
    h = (h * _M(0xc2b2ae35)).astype(np.uint32)
    h ^= h >> _M(16)
    return h


def rand_u32(seed, stream, row, col):
    with np.errstate(over='ignore'):
        seed, stream = np.uint32(seed), np.uint32(stream)
        base = fmix32(np.asarray(seed ^ np.uint32((int(stream) * 0x9e3779b1) & 0xffffffff), dtype=np.uint32))
        h = fmix32(base + row.astype(np.uint32) * _M(0x85ebca77) + col.astype(np.uint32) * _M(0xc2b2ae3d))
    return h


def _keep(seed, stream, rows, cols, p):
    """keep-scale matrix (len(rows), len(cols)): 1/(1-p) where kept, 0 where dropped"""
    thresh = int(p * 65536.0 + 0.5)
    r = np.asarray(rows, dtype=np.uint32)[:, None]
    c = np.asarray(cols, dtype=np.uint32)[None, :]
    h = rand_u32(seed, stream, np.broadcast_to(r, (r.shape[0], c.shape[1])), np.broadcast_to(c >> _M(1), (r.shape[0], c.shape[1])))
    r16 = np.where((c & _M(1)) == 1, h >> _M(16), h & _M(0xffff))
    return torch.from_numpy(np.where(r16 >= thresh, np.float32(1.0 / (1.0 - p)), np.float32(0.0)))


def attn_dropout_mask(seed, stream_id, B, H, N, p):
    """(B, H, N, N) mask of e2k_attn_fwd(p_drop=p, seed, stream_id): element [b,h,q,key]"""
    out = torch.empty(B, H, N, N)
    idx = np.arange(N)
    for b in range(B):
        for h in range(H):
            out[b, h] = _keep(seed, (stream_id * 8192 + b * H + h) & 0xffffffff, idx, idx, p)
    return out


def geglu_dropout_mask(seed, stream_id, M, F, p):
    """(M, F) mask of e2k_geglu_fwd(p_drop=p, seed, stream_id)"""
    return _keep(seed, stream_id, np.arange(M), np.arange(F), p)
