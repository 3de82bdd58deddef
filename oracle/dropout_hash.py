"""TEST INFRASTRUCTURE (oracle side): bit-exact numpy restatement of the counter-hash dropout masks the HIP kernels
generate (e2k_device.h: fmix32 / rand_base / rand_at; attn.hip: drop4 / drop_sample; elementwise.hip: keep_scale).

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
    """keep-scale matrix (len(rows), len(cols)) of the GEGLU kernel (elementwise.hip keep_scale): one hash per PAIR of
    columns, low / high 16 bits; 1/(1-p) where kept, 0 where dropped"""
    thresh = int(p * 65536.0 + 0.5)
    r = np.asarray(rows, dtype=np.uint32)[:, None]
    c = np.asarray(cols, dtype=np.uint32)[None, :]
    h = rand_u32(seed, stream, np.broadcast_to(r, (r.shape[0], c.shape[1])), np.broadcast_to(c >> _M(1), (r.shape[0], c.shape[1])))
    r16 = np.where((c & _M(1)) == 1, h >> _M(16), h & _M(0xffff))
    return torch.from_numpy(np.where(r16 >= thresh, np.float32(1.0 / (1.0 - p)), np.float32(0.0)))


def _mad24(x, y, z):
    """v_mad_u32_u24: (x & 0xffffff) * (y & 0xffffff) + z, low 32 bits"""
    return ((x.astype(np.uint64) & np.uint64(0xffffff)) * np.uint64(y & 0xffffff) + z.astype(np.uint64)).astype(np.uint32)


def drop4_words(h):
    """attn.hip drop4(): the two 32-bit words (four 16-bit samples) of the counter value h"""
    h = h.astype(np.uint32)
    x = h ^ (h >> _M(16))
    x = _mad24(x, 0x85ebcb, h >> _M(8))
    x = x ^ (x >> _M(13))
    a = _mad24(x, 0xc2b2af, x >> _M(11))
    a = a ^ (a >> _M(15))
    b = _mad24(a, 0x9e3779, x >> _M(7))
    b = b ^ (b >> _M(12))
    return a, b


def _keep4(seed, stream, rows, cols, p):
    """keep-scale matrix of the attention kernels (attn.hip drop4 / drop_sample): one hash per group of FOUR consecutive
    keys of the counter h = rand_base(seed, stream) + q * 0x85ebca77 + (key >> 2) * 0xc2b2ae3d -> two 32-bit words; key & 3
    selects (word 0 low, word 0 high, word 1 low, word 1 high) 16 bits; kept iff that sample >= thresh"""
    thresh = int(p * 65536.0 + 0.5)
    r = np.asarray(rows, dtype=np.uint32)[:, None]
    c = np.asarray(cols, dtype=np.uint32)[None, :]
    shape = (r.shape[0], c.shape[1])
    with np.errstate(over='ignore'):
        seed, stream = np.uint32(seed), np.uint32(stream)
        base = fmix32(np.asarray(seed ^ np.uint32((int(stream) * 0x9e3779b1) & 0xffffffff), dtype=np.uint32))
        h = (base + np.broadcast_to(r, shape) * _M(0x85ebca77) + np.broadcast_to(c >> _M(2), shape) * _M(0xc2b2ae3d)).astype(np.uint32)
    w0, w1 = drop4_words(h)
    j = np.broadcast_to(c & _M(3), shape)
    w = np.where((j & _M(2)) != 0, w1, w0)
    r16 = np.where((j & _M(1)) != 0, w >> _M(16), w & _M(0xffff))
    return torch.from_numpy(np.where(r16 >= thresh, np.float32(1.0 / (1.0 - p)), np.float32(0.0)))


def attn_stream(stream_id, bh):
    """attn.hip attn_stream(): top bit set so that attention streams never meet the GEGLU streams (plain call ids)"""
    return (0x80000000 | ((int(stream_id) << 16) & 0x7fff0000) | int(bh)) & 0xffffffff


def _keep4_square(seed, stream, N, p):
    """_keep4 for rows = cols = arange(N), hashing each (query, group of four keys) counter ONCE (the general form above
    hashes it four times): the (N, N) keep-scale matrix, bit-identical to _keep4(seed, stream, arange(N), arange(N), p)"""
    thresh = int(p * 65536.0 + 0.5)
    n4 = (N + 3) // 4
    r = np.arange(N, dtype=np.uint32)[:, None]
    c4 = np.arange(n4, dtype=np.uint32)[None, :]
    with np.errstate(over='ignore'):
        seed, stream = np.uint32(seed), np.uint32(stream)
        base = fmix32(np.asarray(seed ^ np.uint32((int(stream) * 0x9e3779b1) & 0xffffffff), dtype=np.uint32))
        h = (base + r * _M(0x85ebca77) + c4 * _M(0xc2b2ae3d)).astype(np.uint32)
    w0, w1 = drop4_words(h)
    r16 = np.stack([w0 & _M(0xffff), w0 >> _M(16), w1 & _M(0xffff), w1 >> _M(16)], axis=-1).reshape(N, 4 * n4)[:, :N]
    return torch.from_numpy(np.where(r16 >= thresh, np.float32(1.0 / (1.0 - p)), np.float32(0.0)))


def attn_dropout_mask(seed, stream_id, B, H, N, p):
    """(B, H, N, N) mask of e2k_attn_fwd(p_drop=p, seed, stream_id): element [b,h,q,key]"""
    out = torch.empty(B, H, N, N)

    def one(bh):
        out[bh // H, bh % H] = _keep4_square(seed, attn_stream(stream_id, bh), N, p)
    if B * H * N * N < (1 << 22):
        for bh in range(B * H):
            one(bh)
    else:                               # (numpy releases the GIL inside its loops: full-size masks in a few threads)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(one, range(B * H)))
    return out


def geglu_dropout_mask(seed, stream_id, M, F, p):
    """(M, F) mask of e2k_geglu_fwd(p_drop=p, seed, stream_id)"""
    return _keep(seed, stream_id, np.arange(M), np.arange(F), p)
