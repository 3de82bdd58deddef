"""Thin torch-tensor front end of the e2k C ABI: argument checking, raw pointers, current HIP stream.

PyTorch is plumbing here (device memory + stream); all arithmetic happens in csrc/*.hip.
"""
from __future__ import annotations

import contextlib

import ctypes

import torch

from . import _lib

bf16 = torch.bfloat16
f32 = torch.float32
E2KError = _lib.E2KError


def host_ok():
    """True only under the test-side host model of the kernels (tests/emu); the product library takes device pointers"""
    return _lib.host_pointers_ok()


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda or _lib.host_pointers_ok()):
            raise _lib.E2KError('e2_tts_pytorch_amd kernels need tensors on a HIP device (no CPU path)')


def _p(t):
    return None if t is None else t.data_ptr()


_pinned_stream = None        # (device index, raw hipStream_t) while a schedule runs under `pinned_stream`


def _stream(t):
    if not t.is_cuda:
        return None
    ps = _pinned_stream
    if ps is not None and ps[0] == t.device.index:
        return ps[1]
    return torch.cuda.current_stream(t.device).cuda_stream


class pinned_stream:
    """Resolve the current HIP stream ONCE for a whole kernel schedule (a forward or backward pass launches ~1300
    kernels on the same stream; `torch.cuda.current_stream()` costs several microseconds of host time per call and the
    eager path is within 15 % of being host bound).  Everything launched inside the block goes to the stream that was
    current on entry -- do not switch streams inside it for e2k ops."""

    def __init__(self, device):
        self.device = torch.device(device)

    def __enter__(self):
        global _pinned_stream
        self.prev = _pinned_stream
        if self.device.type == 'cuda':
            _pinned_stream = (self.device.index if self.device.index is not None else torch.cuda.current_device(),
                              torch.cuda.current_stream(self.device).cuda_stream)
        return self

    def __exit__(self, *exc):
        global _pinned_stream
        _pinned_stream = self.prev
        return False


def lib():
    return _lib.get()


def raw_stream(device):
    """hipStream_t of torch's current stream on `device` (None on the host model)"""
    device = torch.device(device)
    if device.type != 'cuda':
        return None
    ps = _pinned_stream
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if ps is not None and ps[0] == idx:
        return ps[1]
    return torch.cuda.current_stream(device).cuda_stream


# ---- launch-plan recording (csrc/plan.h): while one is active the GEMM / attention wrappers note their algorithmic
# FLOPs next to the index of the recorded call, so that a profiled replay can be priced against the roofline
_rec_meta = None


def begin_recording(meta=None):
    global _rec_meta
    _lib.get().e2k_plan_begin()
    _rec_meta = meta


def recorded():
    return _lib.get().e2k_query_plan_recorded()


def end_recording():
    global _rec_meta
    _rec_meta = None
    h = _lib.get().e2k_query_plan_end()
    if h <= 0:
        raise _lib.E2KError(f'e2k_query_plan_end failed with code {h}')
    return h


def abort_recording():
    global _rec_meta
    _rec_meta = None
    _lib.get().e2k_plan_abort()


def _note(flops):
    m = _rec_meta
    if m is not None:
        m.append((_lib.get().e2k_query_plan_recorded(), float(flops)))


def profile_plan(handle, meta, phase, device):
    """replay `handle` once with a HIP event after every call -> [dict(name, ms, flops, phase)] (synchronises)"""
    import ctypes
    L = _lib.get()
    n = L.e2k_query_plan_size(handle)
    ms = (ctypes.c_float * max(n, 1))()
    L.e2k_plan_profile(handle, 0, n, ctypes.addressof(ms), raw_stream(device))
    buf = ctypes.create_string_buffer(64)
    flops = dict(meta or ())
    out = []
    for i in range(n):
        L.e2k_plan_op_name(handle, i, ctypes.addressof(buf), 64)
        out.append(dict(name=buf.value.decode(), ms=float(ms[i]), flops=flops.get(i, 0.0), phase=phase, index=i, lane=L.e2k_query_plan_op_lane(handle, i)))
    return out


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
        assert colscale.dtype == f32 and colscale.dim() == 2 and colscale.stride(1) == 1 and colscale.shape[1] == N and rows_per_batch > 0
    if rowmask is not None:
        assert rowmask.dtype in (torch.uint8, torch.bool) and rowmask.numel() == M and rowmask.is_contiguous()
    ldr = 0
    if resid is not None:
        assert resid.dtype == bf16 and resid.shape == (M, N) and resid.stride(1) == 1
        ldr = resid.stride(0)
    if _gemm_shapes is not None:
        key = (M, N, K1, K2, int(out.dtype == f32), int(resid is not None))
        _gemm_shapes[key] = _gemm_shapes.get(key, 0) + 1
    prof = _gemm_profile
    if prof is not None and a.is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _note(2.0 * M * N * (K1 + K2))
    stream = _stream(a)
    _lib.get().e2k_gemm_nt_bf16(_p(a), lda, K1, _p(a2), lda2, K2, _p(b), ldb, _p(out), out.stride(0),
                                int(out.dtype == f32), int(accumulate), M, N, _p(bias), _p(colscale),
                                0 if colscale is None else colscale.stride(0), int(rows_per_batch), _p(rowmask), _p(resid), ldr,
                                gemm_flags, *_nt_ws(a.device, stream), stream)
    if prof is not None and a.is_cuda:
        e1.record()
        prof.append((2.0 * M * N * (K1 + K2), e0, e1))
    return out


def gemm_nt2(a, b, nsplit, *, a2=None, resid=None, resid2=None):
    """(out1 (M, nsplit), out2 (M, N - nsplit)) = split of [a|a2] @ b.T (+ resid / resid2) over the output columns, one launch
    (e2k_gemm_nt2_bf16); nsplit a multiple of 256"""
    _chk(a, b, a2, resid, resid2)
    assert a.dtype == bf16 and b.dtype == bf16
    M, lda = _rows(a)
    N, ldb = _rows(b)
    K1 = a.shape[1]
    K2, lda2 = 0, 0
    if a2 is not None:
        assert a2.dtype == bf16 and a2.shape[0] == M
        _, lda2 = _rows(a2)
        K2 = a2.shape[1]
    assert b.shape[1] == K1 + K2 and 0 < nsplit < N and nsplit % 256 == 0
    out1 = torch.empty((M, nsplit), dtype=bf16, device=a.device)
    out2 = torch.empty((M, N - nsplit), dtype=bf16, device=a.device)
    assert (resid is None) == (resid2 is None)
    if resid is not None:
        assert resid.dtype == bf16 and resid.shape == out1.shape and resid.stride(1) == 1
        assert resid2.dtype == bf16 and resid2.shape == out2.shape and resid2.stride(1) == 1
    if _gemm_shapes is not None:
        key = (M, N, K1, K2, 0, int(resid is not None))
        _gemm_shapes[key] = _gemm_shapes.get(key, 0) + 1
    _note(2.0 * M * N * (K1 + K2))
    stream = _stream(a)
    _lib.get().e2k_gemm_nt2_bf16(_p(a), lda, K1, _p(a2), lda2, K2, _p(b), ldb, M, N, int(nsplit), _p(out1), out1.stride(0),
                                 _p(out2), out2.stride(0), _p(resid), 0 if resid is None else resid.stride(0),
                                 _p(resid2), 0 if resid2 is None else resid2.stride(0), gemm_flags, *_nt_ws(a.device, stream), stream)
    return out1, out2


_nt_ws_cache = {}


def _nt_ws(device, stream):
    """fp32 scratch of the NT GEMM remainder split: one per (device, stream) -- launches of one stream use it one after
    the other, launches of different streams (launch lanes) may overlap."""
    key = (device, stream)
    w = _nt_ws_cache.get(key)
    if w is None:
        w = torch.empty(_lib.get().e2k_query_gemm_nt_ws_bytes() // 4, dtype=f32, device=device)
        _nt_ws_cache[key] = w
    return _p(w), w.numel() * 4


_gemm_profile = None
_gemm_shapes = None        # tools/nt_shapes.py: dict counting the (M, N, K1, K2, out_f32, has_resid) of every gemm_nt call
# E2K_GEMM_* bits passed to every gemm_nt call (A/B benchmarking of kernel variants: 64 = 256 x 128 tile, 128 = 256 x 256
# 8-phase tile for every shape, 256 = the same where it fills the chip).  E2K_GEMM_FLAGS in the environment presets it.
# ---- launch lanes (csrc/plan.h) --------------------------------------------------------------------------------------
MAIN, TEXT, WGRAD = 0, 1, 2


class Lanes:
    """The backbone's schedule on up to three launch lanes: MAIN (the caller's stream), TEXT (the text stream's branches of
    layer i + 1 run next to the audio branches of layer i: both only need the cross projection of layer i) and WGRAD (no
    weight gradient is read before the optimizer / the gradient all-reduce).  The same calls drive the eager schedule
    (torch streams + events, executed now) and a launch-plan recording (e2k_plan_lane / e2k_plan_event_*: what the replay
    will do); on the host model of the kernels only the recording half exists.  Event ids are local to one Lanes object =
    one recording."""

    def __init__(self, device, side_streams, mask=3):
        self.dev = torch.device(device)
        self.mask = mask              # bit k - 1: lane k in use (A/B and fault isolation)
        self.on = len(side_streams) > 0 and mask != 0
        self.cuda = self.on and self.dev.type == 'cuda'
        self.cur = MAIN
        self.n = 1 + len(side_streams)
        self.streams = ([torch.cuda.current_stream(self.dev)] + list(side_streams)) if self.cuda else None
        self._ev = []

    def has(self, k):
        return self.on and k < self.n and (k == 0 or bool(self.mask >> (k - 1) & 1))

    @contextlib.contextmanager
    def lane(self, k):
        """launch what the block launches on lane k (allocations made inside belong to that lane's stream)"""
        if not self.has(k) or k == self.cur:
            yield
            return
        prev, self.cur = self.cur, k
        L = _lib.get()
        L.e2k_plan_lane(k)
        try:
            if self.cuda:
                with torch.cuda.stream(self.streams[k]), pinned_stream(self.dev):
                    yield
            else:
                yield
        finally:
            self.cur = prev
            L.e2k_plan_lane(prev)

    def record(self, k=None):
        """ordering point: everything launched on lane k so far; returns the event id"""
        if not self.on:
            return -1
        k = self.cur if k is None else k
        e = len(self._ev)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(self.streams[k])
            self._ev.append(ev)
        else:
            self._ev.append(None)
        _lib.get().e2k_plan_event_record(k, e)
        return e

    def wait(self, k, e):
        """lane k does not start anything new before event e has happened"""
        if not self.on or e < 0:
            return
        if self.cuda:
            self.streams[k].wait_event(self._ev[e])
        _lib.get().e2k_plan_event_wait(k, e)

    def fence(self, frm, to):
        if self.has(frm) and self.has(to) and frm != to:
            self.wait(to, self.record(frm))

    def join(self):
        """MAIN waits for every side lane (end of a pass)"""
        for k in range(1, self.n):
            self.fence(k, MAIN)


_hip_rt = None


def cu_masked_stream(device, first_cu, n_cus, total=256):
    """a HIP stream whose kernels may only run on `n_cus` of the chip's CUs, [first_cu, first_cu + n_cus) in the runtime's CU
    numbering (hipExtStreamCreateWithCUMask; the driver deals consecutive mask bits round-robin over the 8 XCDs, so a run of
    64 bits is 8 CUs of every XCD), wrapped for torch.  A/B instrument for the launch lanes."""
    global _hip_rt
    import re
    if _hip_rt is None:
        path = None
        for line in open('/proc/self/maps'):
            m = re.search(r'(/\S*libamdhip64\.so\S*)', line)
            if m:
                path = m.group(1)
                break
        _hip_rt = ctypes.CDLL(path or 'libamdhip64.so')
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in range(first_cu, min(first_cu + n_cus, total)):
        mask[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    dev = torch.device(device)
    with torch.cuda.device(dev):
        rc = _hip_rt.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise E2KError(f'hipExtStreamCreateWithCUMask failed with {rc}')
    return torch.cuda.ExternalStream(st.value, device=dev)


def run_plan(handle, first, count, device, side_streams=()):
    """replay calls [first, first + count) of a plan: lane 0 on the current stream, lanes 1.. on `side_streams`"""
    import ctypes
    L = _lib.get()
    dev = torch.device(device)
    if not side_streams:
        return L.e2k_plan_run(handle, first, count, raw_stream(dev))
    n = 1 + len(side_streams)
    arr = (ctypes.c_void_p * n)()
    arr[0] = raw_stream(dev)
    for i, ss in enumerate(side_streams):
        arr[1 + i] = ss.cuda_stream if dev.type == 'cuda' else None
    return L.e2k_plan_run_lanes(handle, first, count, ctypes.addressof(arr), n)


def _stream_array(dev, side_streams):
    import ctypes
    n = 1 + len(side_streams)
    arr = (ctypes.c_void_p * n)()
    arr[0] = raw_stream(dev)
    for i, ss in enumerate(side_streams):
        arr[1 + i] = ss.cuda_stream if dev.type == 'cuda' else None
    return arr, n


def capture_graph(handle, first, count, device, side_streams=()):
    """calls [first, first + count) of a plan as ONE HIP graph (e2k_query_plan_graph_capture): captured on the current stream with the
    lanes on `side_streams`; nothing executes.  -> graph handle for launch_graph"""
    import ctypes
    dev = torch.device(device)
    arr, n = _stream_array(dev, side_streams or ())
    g = _lib.get().e2k_query_plan_graph_capture(handle, first, count, ctypes.addressof(arr), n)
    if g <= 0:
        raise _lib.E2KError(f'e2k_query_plan_graph_capture failed with code {g}')
    return g


def launch_graph(graph, device):
    return _lib.get().e2k_plan_graph_launch(graph, raw_stream(torch.device(device)))


import os as _os
gemm_flags = int(_os.environ.get('E2K_GEMM_FLAGS', '0'))


def set_gemm_profile(lst):
    """bench.py: when a list is installed, every gemm_nt launch is bracketed by HIP events on its stream and
    (flops, start, end) is appended -- used for the live roofline figure."""
    global _gemm_profile
    _gemm_profile = lst


# weight-gradient kernel choice handed to every gemm_tn call (1 = the library chooses per shape, 2 = always 128 x 128,
# 3 = 256 x 256 8-phase kernel wherever it can run); E2K_TN_MODE in the environment presets it (A/B benchmarking)
tn_mode = int(_os.environ.get('E2K_TN_MODE', '1'))


def gemm_tn(a, b, out, *, splits=0, use_tr=True, colsum=None, colsum_from=0, hold=None):
    """out[N,K] += a[M,N].T @ b[M,K]   (fp32 out, bf16 a/b); optionally colsum[n] += sum_m a[m][n] for n >= colsum_from"""
    _chk(a, b, out)
    assert a.dtype == bf16 and b.dtype == bf16 and out.dtype == f32
    M, lda = _rows(a)
    M2, ldb = _rows(b)
    assert M == M2
    N, K = a.shape[1], b.shape[1]
    assert out.shape == (N, K) and out.stride(1) == 1
    lib = _lib.get()
    mode = tn_mode if use_tr is True else int(use_tr)
    nws = lib.e2k_query_gemm_tn_ws_floats(M, N, K, int(splits), mode)
    ws = torch.empty((nws,), dtype=f32, device=a.device) if nws > 0 else None
    if colsum is not None:
        _chk(colsum)
        assert colsum.dtype == f32 and colsum.numel() == N and colsum.is_contiguous()
    _note(2.0 * M * N * K)
    stream = _stream(a)
    lib.e2k_gemm_tn_bf16(_p(a), lda, _p(b), ldb, _p(out), out.stride(0), M, N, K, int(splits), mode, _p(ws),
                         _p(colsum), int(colsum_from), stream)
    if hold is not None:              # launched on a side lane: the caller keeps the operands alive until that lane has been waited for
        hold.append((a, b, ws))
    return out


def can_gemm_tn_dual(M, N1, K1):
    """shapes e2k_gemm_tn_dual_bf16 takes: block boundaries on 256-column tile boundaries, 64-row reduction steps"""
    return M % 64 == 0 and N1 % 256 == 0 and K1 % 256 == 0


def gemm_tn_dual(a1, a2, b1, b2, out, *, splits=0, hold=None):
    """out[N1+N2, K1+K2] += cat(a1, a2, 1).T @ cat(b1, b2, 1) in one launch, neither concatenation materialised
    (a2 / b2 may be None).  fp32 out, bf16 operands."""
    _chk(a1, a2, b1, b2, out)
    M, lda1 = _rows(a1)
    N1, K1 = a1.shape[1], b1.shape[1]
    _, ldb1 = _rows(b1)
    N2 = a2.shape[1] if a2 is not None else 0
    K2 = b2.shape[1] if b2 is not None else 0
    lda2 = _rows(a2)[1] if a2 is not None else 0
    ldb2 = _rows(b2)[1] if b2 is not None else 0
    assert out.dtype == f32 and out.shape == (N1 + N2, K1 + K2) and out.stride(1) == 1
    lib = _lib.get()
    nws = lib.e2k_query_gemm_tn_ws_floats(M, N1 + N2, K1 + K2, int(splits), 3)
    ws = torch.empty((nws,), dtype=f32, device=a1.device) if nws > 0 else None
    _note(2.0 * M * (N1 + N2) * (K1 + K2))
    lib.e2k_gemm_tn_dual_bf16(_p(a1), lda1, N1, _p(a2), lda2, N2, _p(b1), ldb1, K1, _p(b2), ldb2, K2, _p(out), out.stride(0),
                              M, int(splits), _p(ws), _stream(a1))
    if hold is not None:
        hold.append((a1, a2, b1, b2, ws))
    return out


class _TNProblem(ctypes.Structure):
    """e2k_tn_problem (include/e2k.h)"""
    _fields_ = [('A', ctypes.c_void_p), ('lda', ctypes.c_int64), ('B', ctypes.c_void_p), ('ldb', ctypes.c_int64),
                ('C', ctypes.c_void_p), ('ldc', ctypes.c_int64), ('N', ctypes.c_int32), ('K', ctypes.c_int32),
                ('colsum', ctypes.c_void_p), ('cs_from', ctypes.c_int32), ('reserved', ctypes.c_int32)]


TN_GROUP_MAX = 8


def can_group_tn(a, b):
    """operands the grouped weight-gradient launch takes: 64-row reduction steps, at least 8 columns each"""
    return a.shape[0] % 64 == 0 and a.shape[1] >= 8 and b.shape[1] >= 8


def gemm_tn_group(problems, *, splits=0, hold=None):
    """problems: [(a (M, N_i) bf16, b (M, K_i) bf16, out (N_i, K_i) fp32, colsum or None, colsum_from)], all with the same
    M: out_i += a_i.T @ b_i (+ colsum_i[n] += sum_m a_i[m][n] for n >= colsum_from) in ONE launch (e2k_gemm_tn_group_bf16)"""
    assert 1 <= len(problems) <= TN_GROUP_MAX
    M = problems[0][0].shape[0]
    arr = (_TNProblem * len(problems))()
    fl = 0.
    for i, (a, b, out, cs, cs_from) in enumerate(problems):
        _chk(a, b, out, cs)
        assert a.dtype == bf16 and b.dtype == bf16 and out.dtype == f32 and a.shape[0] == M and b.shape[0] == M
        N, K = a.shape[1], b.shape[1]
        assert out.shape == (N, K) and out.stride(1) == 1
        if cs is not None:
            assert cs.dtype == f32 and cs.numel() == N and cs.is_contiguous()
        arr[i] = _TNProblem(_p(a), _rows(a)[1], _p(b), _rows(b)[1], _p(out), out.stride(0), N, K, _p(cs), int(cs_from), 0)
        fl += 2.0 * M * N * K
    lib = _lib.get()
    ap = ctypes.cast(arr, ctypes.c_void_p)
    nws = lib.e2k_query_gemm_tn_group_ws_floats(ap, len(problems), M, int(splits))
    ws = torch.empty((nws,), dtype=f32, device=problems[0][0].device) if nws > 0 else None
    _note(fl)
    lib.e2k_gemm_tn_group_bf16(ap, len(problems), M, int(splits), _p(ws), _stream(problems[0][0]))
    if hold is not None:
        hold.append((problems, ws))


# ------------------------------------------------------------------------------------------------ hyper-connections

HC_PARAM_NAMES = ('static_beta', 'static_alpha', 'dynamic_alpha_fn', 'dynamic_alpha_scale', 'dynamic_beta_fn',
                  'dynamic_beta_scale', 'norm.gamma')


def hc_coef_width():
    return _lib.get().e2k_query_hc_coef_width()


def hc_fwd(xin, params, *, yprev=None, coef_prev=None, width=True, norm=None, want_bin=True):
    """xin (Mtok,4,D) bf16.  Returns (Mout, bin, coef) if width else (X, None, None).
    norm = (gamma fp32 (nb, D), gamma_off, rows_per_batch): the branch's RMSNorm in the same launch (e2k_hc_fwd_norm) ->
    (Mout, bin or None, coef, xn, rn); want_bin=False: the un-normalised branch input is not written (no-grad forwards)"""
    _chk(xin, yprev, coef_prev)
    Mtok, S, D = xin.shape
    assert S == 4 and xin.dtype == bf16 and xin.is_contiguous()
    depth = yprev is not None
    mout = torch.empty_like(xin)
    binp = coef = None
    if width:
        binp = torch.empty((Mtok, D), dtype=bf16, device=xin.device) if (want_bin or norm is None) else None
        coef = torch.empty((Mtok, hc_coef_width()), dtype=f32, device=xin.device)
    if norm is not None:
        assert width
        gamma, goff, rpb = norm
        _chk(gamma)
        assert gamma.dtype == f32 and gamma.dim() == 2 and gamma.stride(1) == 1 and gamma.shape[1] == D
        xn = torch.empty((Mtok, D), dtype=bf16, device=xin.device)
        rn = torch.empty((Mtok,), dtype=f32, device=xin.device) if want_bin else None
        _lib.get().e2k_hc_fwd_norm(_p(xin), _p(yprev), _p(coef_prev), _p(mout), _p(binp), _p(coef), *[_p(t) for t in params], Mtok, D,
                                   int(depth), _p(gamma), gamma.stride(0), float(goff), int(rpb), _p(xn), _p(rn), _stream(xin))
        return mout, binp, coef, xn, rn
    ps = [_p(t) for t in params] if width else [None] * 7
    _lib.get().e2k_hc_fwd(_p(xin), _p(yprev), _p(coef_prev), _p(mout), _p(binp), _p(coef), *ps, Mtok, D,
                          int(depth), int(width), _stream(xin))
    return mout, binp, coef


def hc_bwd(G, *, xin=None, yprev=None, coef_prev=None, dbin=None, ycur=None, coef=None, params=None, grads=None, deferred=None):
    """Backward of hc_fwd. width iff coef is given; depth iff yprev is given.
    Returns (dR, dyprev); with width=False dR is G itself.  deferred (a list): the reduction of the per-workgroup partials
    into the parameter gradients is NOT launched; a zero-argument callable that launches it (on whatever stream is current
    then) is appended instead"""
    _chk(G, xin, yprev, coef_prev, dbin, ycur, coef)
    Mtok, S, D = G.shape
    width, depth = coef is not None, yprev is not None
    lib = _lib.get()
    dR = torch.empty_like(G) if width else G
    dyprev = torch.empty((Mtok, D), dtype=bf16, device=G.device) if depth else None
    partial = None
    if width:
        partial = torch.empty((lib.e2k_query_hc_bwd_blocks(Mtok, D), lib.e2k_query_hc_partial_stride(D)), dtype=f32,
                              device=G.device)
    ps = [_p(t) for t in params] if width else [None] * 7
    gs = [_p(t) for t in grads] if width else [None] * 7
    lib.e2k_hc_bwd(_p(xin), _p(yprev), _p(coef_prev), _p(G), _p(dbin), _p(ycur), _p(coef),
                   _p(dR) if width else None, _p(dyprev), *ps, *gs, _p(partial), Mtok, D, int(depth),
                   (2 if deferred is not None else 1) if width else 0, _stream(G))
    if width and deferred is not None:
        deferred.append(HCReduce(partial, params, grads, Mtok, D))
    return dR, dyprev


class _HCReduceItem(ctypes.Structure):
    """e2k_hc_reduce_item (include/e2k.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ('partial', 'dyn_alpha_fn', 'dyn_beta_fn', 'gamma', 'g_static_beta', 'g_static_alpha',
                                               'g_dyn_alpha_fn', 'g_dyn_alpha_scale', 'g_dyn_beta_fn', 'g_dyn_beta_scale', 'g_gamma')] + \
               [('Mtok', ctypes.c_int32), ('D', ctypes.c_int32)]


HC_BATCH_MAX = 8


class HCReduce:
    """the deferred second half of an hc_bwd(width) call: per-workgroup partials -> parameter gradients.  Calling it launches
    it alone (on whatever stream is current); `launch_hc_reduces` launches up to HC_BATCH_MAX of them as one kernel.  The
    object keeps the partial buffer alive until it is dropped."""

    def __init__(self, partial, params, grads, Mtok, D):
        self.partial, self.params, self.grads, self.Mtok, self.D = partial, params, grads, Mtok, D

    def item(self):
        return _HCReduceItem(_p(self.partial), _p(self.params[2]), _p(self.params[4]), _p(self.params[6]),
                             *[_p(t) for t in self.grads], self.Mtok, self.D)

    def __call__(self):
        _lib.get().e2k_hc_bwd_reduce(_p(self.partial), _p(self.params[2]), _p(self.params[4]), _p(self.params[6]),
                                     *[_p(t) for t in self.grads], self.Mtok, self.D, _stream(self.partial))


def launch_hc_reduces(items):
    """items: [HCReduce]; one launch per HC_BATCH_MAX of them (e2k_hc_bwd_reduce_batch)"""
    for i in range(0, len(items), HC_BATCH_MAX):
        chunk = items[i:i + HC_BATCH_MAX]
        if len(chunk) == 1:
            chunk[0]()
            continue
        arr = (_HCReduceItem * len(chunk))(*[c.item() for c in chunk])
        _lib.get().e2k_hc_bwd_reduce_batch(ctypes.cast(arr, ctypes.c_void_p), len(chunk), _stream(chunk[0].partial))


# ------------------------------------------------------------------------------------------------ norms / gates / GEGLU

def rmsnorm_fwd(x, gamma, gamma_off, rows_per_batch):
    """x (M,D) bf16; gamma fp32 (nb,D).  -> (y bf16, rn fp32 (M,))"""
    _chk(x, gamma)
    M, D = x.shape
    assert x.dtype == bf16 and x.is_contiguous() and gamma.dtype == f32 and gamma.dim() == 2 and gamma.stride(1) == 1
    y = torch.empty_like(x)
    rn = torch.empty((M,), dtype=f32, device=x.device)
    _lib.get().e2k_rmsnorm_fwd(_p(x), _p(gamma), gamma.stride(0), float(gamma_off), int(rows_per_batch), _p(y), _p(rn),
                               M, D, _stream(x))
    return y, rn


def rmsnorm_bwd(dy, x, rn, gamma, gamma_off, rows_per_batch, dgamma):
    _chk(dy, x, rn, gamma, dgamma)
    M, D = x.shape
    assert dy.dtype == bf16 and dy.is_contiguous() and dgamma.dtype == f32 and dgamma.stride() == gamma.stride()
    dx = torch.empty_like(x)
    _lib.get().e2k_rmsnorm_bwd(_p(dy), _p(x), _p(rn), _p(gamma), gamma.stride(0), float(gamma_off), int(rows_per_batch),
                               _p(dx), _p(dgamma), M, D, _stream(x))
    return dx


def gate_bwd(dy, y, g, gsum, rows_per_batch):
    _chk(dy, y, g, gsum)
    M, D = y.shape
    assert dy.is_contiguous() and y.is_contiguous() and g.dtype == f32 and gsum.dtype == f32 and g.stride() == gsum.stride()
    dao = torch.empty_like(dy)
    _lib.get().e2k_gate_bwd(_p(dy), _p(y), _p(g), _p(dao), _p(gsum), g.stride(0), M, D, int(rows_per_batch), _stream(y))
    return dao


def geglu_fwd(H, p_drop=0., seed=0, stream_id=0, seed_dev=None):
    _chk(H)
    M, F2 = H.shape
    assert H.dtype == bf16 and H.stride(1) == 1
    out = torch.empty((M, F2 // 2), dtype=bf16, device=H.device)
    _lib.get().e2k_geglu_fwd(_p(H), H.stride(0), _p(out), M, F2 // 2, float(p_drop), int(seed), _p(seed_dev), int(stream_id),
                             _stream(H))
    return out


# GEGLU as the epilogue of FeedForward's first GEMM in no-grad forwards (E2K_FUSE_GEGLU=0: always two launches)
fuse_geglu = bool(int(_os.environ.get('E2K_FUSE_GEGLU', '1')))


def can_fuse_geglu(M, F, K):
    return bool(_lib.get().e2k_query_gemm_nt_geglu(int(M), int(F), int(K)))


def gemm_nt_geglu(a, w1, bias=None, p_drop=0., seed=0, stream_id=0, seed_dev=None, want_h=True):
    """(H, act): H (M, 2F) = a @ w1.T + bias (None if not want_h), act (M, F) = H[:, :F] * gelu(H[:, F:]) * keep --
    one launch; act is geglu_fwd of the returned H (erf to 1.5e-7).  Shapes: can_fuse_geglu(M, F, K)."""
    _chk(a, w1, bias, seed_dev)
    assert a.dtype == bf16 and w1.dtype == bf16
    M, lda = _rows(a)
    N, ldb = _rows(w1)
    K, F = a.shape[1], N // 2
    assert w1.shape[1] == K and N == 2 * F
    if bias is not None:
        assert bias.dtype == f32 and bias.numel() == N and bias.is_contiguous()
    H = torch.empty((M, N), dtype=bf16, device=a.device) if want_h else None
    act = torch.empty((M, F), dtype=bf16, device=a.device)
    _note(2.0 * M * N * K)
    stream = _stream(a)
    _lib.get().e2k_gemm_nt_geglu_bf16(_p(a), lda, K, _p(w1), ldb, _p(bias), _p(H), 0 if H is None else H.stride(0), _p(act), act.stride(0),
                                      M, F, float(p_drop), int(seed), _p(seed_dev), int(stream_id), gemm_flags,
                                      *_nt_ws(a.device, stream), stream)
    return H, act


# GEGLU backward as the epilogue of FeedForward's second dgrad GEMM (E2K_FUSE_GEGLU_BWD=0: two launches)
fuse_geglu_bwd = bool(int(_os.environ.get('E2K_FUSE_GEGLU_BWD', '1')))
# remainder split (E2K_GEMM_SPLIT) for this launch only: its tiles end in a long element-wise epilogue, so the 16 tiles that 528 leave over 512 slots
# cost a whole extra round (A/B instrument, round 6)
geglu_bwd_split = int(_os.environ.get('E2K_GEGLU_BWD_SPLIT', '0'))


def can_fuse_geglu_bwd(M, F, K):
    """should FeedForward's backward take the fused launch?  Not where the library itself would not pick its 256 x 256 kernel for
    this output (few tiles), nor under the A/B flags that take that kernel or its LDS-DMA staging away (E2K_GEMM_NO_T256 / NO_GLDS)"""
    if gemm_flags & (1 | 256):
        return False
    return _lib.get().e2k_query_gemm_nt_geglu_bwd(int(M), int(F), int(K)) == 1


def gemm_nt_geglu_bwd(dy, w2T, H, p_drop=0., seed=0, stream_id=0, seed_dev=None):
    """dH (M, 2F) = geglu_bwd(dy @ w2T.T, H) in one launch (d(act) is never written); dy (M, K), w2T (F, K), H (M, 2F).
    Shapes: can_fuse_geglu_bwd(M, F, K)."""
    _chk(dy, w2T, H, seed_dev)
    assert dy.dtype == bf16 and w2T.dtype == bf16 and H.dtype == bf16
    M, ldy = _rows(dy)
    F, ldb = _rows(w2T)
    K = dy.shape[1]
    assert w2T.shape[1] == K and H.shape == (M, 2 * F) and H.stride(1) == 1
    dH = torch.empty((M, 2 * F), dtype=bf16, device=dy.device)
    if _gemm_shapes is not None:          # (tools/nt_shapes.py: counted as the plain NT GEMM of its shape)
        key = (M, F, K, 0, 0, 0)
        _gemm_shapes[key] = _gemm_shapes.get(key, 0) + 1
    _note(2.0 * M * F * K)
    stream = _stream(dy)
    _lib.get().e2k_gemm_nt_geglu_bwd_bf16(_p(dy), ldy, K, _p(w2T), ldb, _p(H), H.stride(0), _p(dH), dH.stride(0), M, F, float(p_drop),
                                          int(seed), _p(seed_dev), int(stream_id), gemm_flags | (512 if geglu_bwd_split else 0), *_nt_ws(dy.device, stream), stream)
    return dH


qk_rot_any_size = False      # tests: take the fused projection wherever the kernel can run, not only where the library recommends it


def can_fuse_qk_rot(M, N, K, H, Npad):
    """can the attention's input projection write q and k rotated and head-major itself (gemm_nt_qkrot)?  Not under the A/B flags that
    take the 256 x 256 kernel, its LDS-DMA staging or its staged epilogue away, nor where the backward kernels ask for transposed
    copies of q / k (rows beyond 4096 or the register-staged family, E2K_ATTN_FLAGS 128: those come out of qkv_post_fwd)"""
    if gemm_flags & (1 | 4 | 8 | 64 | 256) or _lib.get().e2k_query_attn_bwd_transposes(int(Npad), attn_probe & 128):
        return False
    q = _lib.get().e2k_query_gemm_nt_qkrot(int(M), int(N), int(K), int(H))
    return q == 1 or (q == 2 and qk_rot_any_size)


def gemm_nt_qkrot(a, w, out, B, H, N, cosb, sinb, bias=None):
    """out[:, 2 H 64:] = (a @ w.T + bias)[:, 2 H 64:]; the q and k columns leave the launch rotated and head-major -> (Q, K), each
    (B, H, N, 64) bf16 (e2k_gemm_nt_qkrot_bf16; qkv_post_fwd(..., qk=(Q, K)) does the rest).  Shapes: can_fuse_qk_rot."""
    _chk(a, w, out, cosb, sinb, bias)
    assert a.dtype == bf16 and w.dtype == bf16 and out.dtype == bf16 and out.stride(1) == 1
    M, lda = _rows(a)
    cols, ldb = _rows(w)
    Kd = a.shape[1]
    assert M == B * N and w.shape[1] == Kd and out.shape == (M, cols)
    if bias is not None:
        assert bias.dtype == f32 and bias.numel() == cols and bias.is_contiguous()
    Q, K = (torch.empty((B, H, N, 64), dtype=bf16, device=a.device) for _ in range(2))
    if _gemm_shapes is not None:          # (tools/nt_shapes.py: counted as the plain NT GEMM of its shape)
        key = (M, cols, Kd, 0, 0, 0)
        _gemm_shapes[key] = _gemm_shapes.get(key, 0) + 1
    _note(2.0 * M * cols * Kd)
    _lib.get().e2k_gemm_nt_qkrot_bf16(_p(a), lda, Kd, _p(w), ldb, _p(bias), _p(out), out.stride(0), _p(Q), _p(K), _p(cosb), _p(sinb),
                                      B, H, N, cols, _stream(a))
    return Q, K


def geglu_bwd(dout, H, p_drop=0., seed=0, stream_id=0, seed_dev=None):
    _chk(dout, H)
    M, F2 = H.shape
    assert dout.is_contiguous() and dout.shape == (M, F2 // 2)
    dH = torch.empty_like(H)
    _lib.get().e2k_geglu_bwd(_p(dout), _p(H), H.stride(0), _p(dH), M, F2 // 2, float(p_drop), int(seed), _p(seed_dev),
                             int(stream_id), _stream(H))
    return dH


def colsum(x, out):
    _chk(x, out)
    M, N = x.shape
    assert x.dtype == bf16 and x.stride(1) == 1 and out.dtype == f32 and out.numel() == N
    _lib.get().e2k_colsum_bf16(_p(x), x.stride(0), _p(out), M, N, _stream(x))
    return out


def cast_bf16(src, dst):
    _chk(src, dst)
    assert src.dtype == f32 and dst.dtype == bf16 and src.numel() == dst.numel()
    _lib.get().e2k_cast_bf16(_p(src), _p(dst), src.numel(), _stream(src))
    return dst


def cast_transpose_bf16(src, dst):
    """src (R,C) fp32 contiguous -> dst (C,R) bf16 (row stride dst.stride(0))"""
    _chk(src, dst)
    R, C = src.shape
    assert src.is_contiguous() and dst.shape == (C, R) and dst.stride(1) == 1
    _lib.get().e2k_cast_transpose_bf16(_p(src), _p(dst), R, C, dst.stride(0), _stream(src))
    return dst


def cast_transpose_batch(flat, flatT, desc, total_blocks):
    """all transposed bf16 shadows in one launch; desc: int64 (n, 6) device tensor, see e2k_cast_transpose_batch"""
    _chk(flat, flatT, desc)
    assert flat.dtype == f32 and flatT.dtype == bf16 and desc.dtype == torch.int64 and desc.is_contiguous() and desc.shape[1] == 6
    _lib.get().e2k_cast_transpose_batch(_p(flat), _p(flatT), _p(desc), desc.shape[0], int(total_blocks), _stream(flat))


# ------------------------------------------------------------------------------------------------ depthwise conv

def dwconv_fwd(x, mask, w, bias, need_pre=True):
    """x (B,N,C) bf16, mask (B,N) bool/u8 or None, w (C,1,ks)/(C,ks) fp32 -> (pre, y); need_pre False: the pre-activation (read by the
    backward pass only) is not written and None is returned for it"""
    _chk(x, mask, w, bias)
    B, N, C = x.shape
    ks = w.shape[-1]
    assert x.is_contiguous() and w.is_contiguous() and w.dtype == f32 and bias.dtype == f32
    pre, y = (torch.empty_like(x) if need_pre else None), torch.empty_like(x)
    _lib.get().e2k_dwconv_fwd(_p(x), _p(mask), _p(w), _p(bias), _p(pre), _p(y), B, N, C, ks, _stream(x))
    return pre, y


# False: the (dw, dbias) partials of the depthwise-conv backward go through global fp32 atomics instead of a workspace +
# reduce pass (A/B; the C ABI accepts ws = NULL)
dwconv_bwd_workspace = True


def dwconv_bwd(dy, pre, x, mask, w, dw, dbias, deferred=None):
    """deferred (a list): the sum of the (dw, dbias) partials is not launched; a callable that launches it is appended"""
    _chk(dy, pre, x, mask, w, dw, dbias)
    B, N, C = x.shape
    ks = w.shape[-1]
    assert dy.is_contiguous() and dw.dtype == f32 and dbias.dtype == f32
    dx = torch.empty_like(x)
    L = _lib.get()
    ws = torch.empty((L.e2k_query_dwconv_bwd_ws_floats(B, N, C, ks),), dtype=f32, device=x.device) if dwconv_bwd_workspace else None
    defer = deferred is not None and ws is not None
    L.e2k_dwconv_bwd(_p(dy), _p(pre), _p(x), _p(mask), _p(w), _p(dx), _p(dw), _p(dbias), _p(ws), B, N, C, ks,
                     1 if defer else 0, _stream(x))
    if defer:
        def reduce_(ws=ws, dw=dw, dbias=dbias):
            L.e2k_dwconv_bwd_reduce(_p(ws), _p(dw), _p(dbias), B, N, C, ks, 1, _stream(ws))
        deferred.append(reduce_)
    return dx


# ------------------------------------------------------------------------------------------------ attention

def rotary_table(n, device, dim_head=64, base=10000.):
    """cos/sin (n, dim_head/2) fp32; x_transformers.RotaryEmbedding.forward_from_seq_len."""
    inv_freq = 1. / (base ** (torch.arange(0, dim_head, 2, dtype=f32, device=device) / dim_head))
    ang = torch.arange(n, dtype=f32, device=device)[:, None] * inv_freq[None, :]
    return ang.cos().contiguous(), ang.sin().contiguous()


class AttnState:
    """buffers produced by qkv_post_fwd / attn_fwd and consumed by the backward"""
    __slots__ = ('Q', 'K', 'V', 'QT', 'KT', 'VT', 'gate', 'mix', 'O', 'Og', 'lse2', 'B', 'H', 'N', 'Npad', 'dropbits', 'laser', 'Vorig')


def qkv_post_fwd(qkvg, B, H, N, cosb, sinb, vfirst=None, laser=0., need_v=True, qk=None):
    """qk = (Q, K): both were written by gemm_nt_qkrot already (the kernel then runs the value path and the gates only).
    laser > 0: LASER attention's value map exp(c tanh(v / c)) (st.laser); on the first layer st.Vorig keeps the values
    before it (the value residual of the later layers).  need_v False: the row-major values are not written (st.V None) -- the
    forward kernels read V^T, only the backward pass and the value residual of the later layers read V"""
    _chk(qkvg, cosb, sinb, vfirst)
    assert qkvg.dtype == bf16 and qkvg.stride(1) == 1 and qkvg.shape[0] == B * N
    dev = qkvg.device
    Npad = (N + 63) // 64 * 64
    st = AttnState()
    st.B, st.H, st.N, st.Npad, st.dropbits = B, H, N, Npad, None
    need = _lib.get().e2k_query_attn_bwd_transposes(Npad, attn_probe & 128)
    if qk is None:
        st.Q, st.K = (torch.empty((B, H, N, 64), dtype=bf16, device=dev) for _ in range(2))
    else:
        st.Q, st.K = qk
        assert not need and st.Q.shape == (B, H, N, 64) and st.K.shape == (B, H, N, 64)
    st.V = torch.empty((B, H, N, 64), dtype=bf16, device=dev) if need_v else None
    st.VT = torch.empty((B, H, 64, Npad), dtype=bf16, device=dev)
    st.KT = torch.empty((B, H, 64, Npad), dtype=bf16, device=dev) if need & 1 else None
    st.QT = torch.empty((B, H, 64, Npad), dtype=bf16, device=dev) if need & 2 else None
    st.gate = torch.empty((B, H, N), dtype=f32, device=dev)
    st.mix = torch.empty((B, H, N), dtype=f32, device=dev) if vfirst is not None else None
    st.laser = float(laser)
    st.Vorig = torch.empty((B, H, N, 64), dtype=bf16, device=dev) if laser > 0 and vfirst is None else None
    _lib.get().e2k_qkv_post_fwd(_p(qkvg), qkvg.stride(0), _p(cosb), _p(sinb), _p(vfirst), _p(st.Q if qk is None else None),
                                _p(st.K if qk is None else None), _p(st.V),
                                _p(st.QT), _p(st.KT), _p(st.VT), _p(st.gate), _p(st.mix), _p(st.Vorig), st.laser, B, H, N, Npad,
                                _stream(qkvg))
    return st


# The forward leaves its dropout keep decisions as wave ballot words (18.9 MB per cfg3 attention) and the two backward
# kernels read them back instead of re-hashing: bit-identical results (tools/attn_share_check.py on MI355X), forward
# 0.132 -> 0.140 ms, backward 0.403 -> 0.358 ms per cfg3 attention.  False = every kernel re-derives the mask.
attn_share_dropmask = True
_ATTN_BOTH = 64 | 128 | 256        # flag bits that select kernel variants (forward and backward calls alike)
attn_probe = int(_os.environ.get('E2K_ATTN_FLAGS', '0'))      # E2K_ATTN_* bits: 128 = the register-staged kernels instead of the LDS-DMA rings (A/B); probes 1..32 give wrong results on purpose


def attn_fwd(st, kmask_pad, p_drop=0., seed=0, stream_id=0, seed_dev=None):
    """kmask_pad (B, Npad) uint8.  Fills st.O / st.Og / st.lse2, returns Og (B*N, H*64)."""
    _chk(kmask_pad)
    B, H, N, Npad = st.B, st.H, st.N, st.Npad
    assert kmask_pad.shape == (B, Npad) and kmask_pad.dtype == torch.uint8 and kmask_pad.is_contiguous()
    dev = st.Q.device
    st.O = torch.empty((B * N, H * 64), dtype=bf16, device=dev)
    st.Og = torch.empty((B * N, H * 64), dtype=bf16, device=dev)
    st.lse2 = torch.empty((B, H, N), dtype=f32, device=dev)
    st.dropbits = None
    if attn_share_dropmask and p_drop > 0:
        nbytes = _lib.get().e2k_query_attn_dropbits_bytes(B, H, N)
        if nbytes > 0:
            st.dropbits = torch.empty(nbytes // 8, dtype=torch.int64, device=dev)
    _note(4.0 * B * H * N * N * 64)
    _lib.get().e2k_attn_fwd(_p(st.Q), _p(st.K), _p(st.VT), _p(kmask_pad), _p(st.gate), _p(st.O), _p(st.Og), _p(st.lse2),
                            _p(st.dropbits), B, H, N, Npad, float(p_drop), int(seed), _p(seed_dev), int(stream_id),
                            attn_probe, _stream(st.Q))
    if st.laser > 0:
        # LASER: the head gates apply to log(out) (x-transformers Attention.forward); st.O keeps the attention's own output
        st.Og = torch.empty_like(st.Og)
        _lib.get().e2k_laser_out_fwd(_p(st.O), _p(st.gate), _p(kmask_pad), _p(st.Og), B, H, N, Npad, _stream(st.Q))
    return st.Og


def attn_bwd(st, dOg, kmask_pad, p_drop=0., seed=0, stream_id=0, seed_dev=None):
    """-> dQ, dK, dV (B,H,N,64) bf16, dgate_pre (B,H,N) fp32"""
    _chk(dOg, kmask_pad)
    B, H, N, Npad = st.B, st.H, st.N, st.Npad
    dev = st.Q.device
    assert dOg.dtype == bf16 and dOg.is_contiguous() and dOg.shape == (B * N, H * 64)
    dO = torch.empty((B, H, N, 64), dtype=bf16, device=dev)
    dOT = torch.empty((B, H, 64, Npad), dtype=bf16, device=dev) if st.QT is not None else None
    delta = torch.empty((B, H, N), dtype=f32, device=dev)
    dgate = torch.empty((B, H, N), dtype=f32, device=dev)
    dQ, dK, dV = (torch.empty((B, H, N, 64), dtype=bf16, device=dev) for _ in range(3))
    dgate_laser = None
    if st.laser > 0:
        dOin, dgate_laser = torch.empty_like(dOg), torch.empty((B, H, N), dtype=f32, device=dev)
        _lib.get().e2k_laser_out_bwd(_p(dOg), _p(st.O), _p(st.gate), _p(kmask_pad), _p(dOin), _p(dgate_laser), B, H, N, Npad,
                                     _stream(dOg))
        dOg = dOin
    _note(10.0 * B * H * N * N * 64)
    _lib.get().e2k_attn_bwd(_p(dOg), _p(st.O), _p(st.gate), _p(st.lse2), _p(st.Q), _p(st.K), _p(st.V), _p(st.QT),
                            _p(st.KT), _p(kmask_pad), _p(st.dropbits), _p(dO), _p(dOT), _p(delta), _p(dgate), _p(dQ), _p(dK),
                            _p(dV), B, H, N, Npad, float(p_drop), int(seed), _p(seed_dev), int(stream_id), attn_probe & _ATTN_BOTH,
                            _stream(dOg))
    return dQ, dK, dV, dgate if dgate_laser is None else dgate_laser


def qkv_post_bwd(st, dQ, dK, dV, dgate_pre, qkvg, cosb, sinb, vfirst=None, dvfirst=None, first_layer=False, out=None):
    """out (tests): a preallocated (M, ld) bf16 buffer to write the gradient rows into"""
    _chk(dQ, dK, dV, dgate_pre, qkvg, vfirst, dvfirst, out)
    B, H, N = st.B, st.H, st.N
    M, cols = qkvg.shape
    ld = qkvg.stride(0)
    if out is not None:
        assert out.shape == (M, ld) and out.dtype == bf16 and out.is_contiguous()
        dqkvg = out[:, :cols]
    elif ld == cols:
        dqkvg = torch.empty((M, cols), dtype=bf16, device=qkvg.device)
    else:                                        # padded row stride: the kernel zeroes the pad columns (they are read as K
        dqkvg = torch.empty((M, ld), dtype=bf16, device=qkvg.device)[:, :cols]      # padding by the dgrad GEMM)
    _lib.get().e2k_qkv_post_bwd(_p(dQ), _p(dK), _p(dV), _p(dgate_pre), _p(qkvg), qkvg.stride(0), _p(cosb), _p(sinb),
                                _p(vfirst), _p(st.mix), _p(dvfirst), int(first_layer), _p(dqkvg), st.laser, B, H, N,
                                _stream(dQ))
    return dqkvg


# ------------------------------------------------------------------------------------------------ glue (csrc/glue.hip)

def fill_(t, value=0):
    """byte fill of a contiguous tensor (hipMemsetAsync as a recordable e2k call); value is a BYTE"""
    _chk(t)
    assert t.is_contiguous()
    _lib.get().e2k_fill_bytes(_p(t), int(value), t.numel() * t.element_size(), _stream(t))
    return t


def fill_cols_(t, c0, value=0):
    """byte fill of columns [c0:] of every row of a 2-D tensor with contiguous rows"""
    _chk(t)
    assert t.dim() == 2 and t.stride(1) == 1
    es = t.element_size()
    _lib.get().e2k_fill_bytes_2d(t.data_ptr() + c0 * es, t.stride(0) * es, int(value), (t.shape[1] - c0) * es, t.shape[0], _stream(t))
    return t


def zeros(shape, dtype, device):
    return fill_(torch.empty(shape, dtype=dtype, device=device))


def cast_f32(src, dst=None):
    _chk(src, dst)
    assert src.dtype == bf16 and src.is_contiguous()
    if dst is None:
        dst = torch.empty(src.shape, dtype=f32, device=src.device)
    assert dst.dtype == f32 and dst.is_contiguous() and dst.numel() == src.numel()
    _lib.get().e2k_cast_f32(_p(src), _p(dst), src.numel(), _stream(src))
    return dst


def sigmoid(src):
    _chk(src)
    assert src.dtype == f32 and src.is_contiguous()
    dst = torch.empty_like(src)
    _lib.get().e2k_sigmoid_f32(_p(src), _p(dst), src.numel(), _stream(src))
    return dst


def build_masks(mask, B, T, R, device, want_mask_n):
    """-> kmask (B, Npad) u8, mask_n (B, N) u8 or None"""
    _chk(mask)
    N = T + R
    Npad = (N + 63) // 64 * 64
    if mask is not None:
        assert mask.shape == (B, T) and mask.is_contiguous() and mask.dtype in (torch.bool, torch.uint8)
    kmask = torch.empty((B, Npad), dtype=torch.uint8, device=device)
    mask_n = torch.empty((B, N), dtype=torch.uint8, device=device) if want_mask_n else None
    _lib.get().e2k_build_masks(_p(mask), _p(kmask), _p(mask_n), B, T, R, Npad, _stream(kmask))
    return kmask, mask_n


def stream_pack_fwd(x, abs_pos, regs):
    """x (B,T,D) fp32, abs_pos (>=T, D) fp32 or None, regs (R,D) fp32 -> X (B*(T+R), 4, D) bf16"""
    _chk(x, abs_pos, regs)
    B, T, D = x.shape
    R = regs.shape[0]
    assert x.dtype == f32 and x.is_contiguous() and regs.dtype == f32 and regs.is_contiguous()
    if abs_pos is not None:
        assert abs_pos.dtype == f32 and abs_pos.is_contiguous() and abs_pos.shape[0] >= T and abs_pos.shape[1] == D
    X = torch.empty((B * (T + R), 4, D), dtype=bf16, device=x.device)
    _lib.get().e2k_stream_pack_fwd(_p(x), _p(abs_pos), _p(regs), _p(X), B, T, R, D, _stream(x))
    return X


def stream_pack_bwd(dX, B, T, R, dregs, dabs):
    """dX (B*(T+R), 4, D) bf16 -> dx (B,T,D) fp32; dregs (R,D) / dabs (T.., D) fp32 accumulated"""
    _chk(dX, dregs, dabs)
    D = dX.shape[-1]
    assert dX.dtype == bf16 and dX.is_contiguous() and dregs.is_contiguous() and (dabs is None or dabs.is_contiguous())
    dx = torch.empty((B, T, D), dtype=f32, device=dX.device)
    _lib.get().e2k_stream_pack_bwd(_p(dX), _p(dx), _p(dregs), _p(dabs), B, T, R, D, _stream(dX))
    return dx


def stream_unpack_fwd(X, B, T, R):
    _chk(X)
    D = X.shape[-1]
    assert X.dtype == bf16 and X.is_contiguous()
    xsum = torch.empty((B * T, D), dtype=bf16, device=X.device)
    _lib.get().e2k_stream_unpack_fwd(_p(X), _p(xsum), B, T, R, D, _stream(X))
    return xsum


def stream_unpack_bwd(dxs, B, T, R):
    _chk(dxs)
    D = dxs.shape[-1]
    assert dxs.dtype == bf16 and dxs.is_contiguous()
    dX = torch.empty((B * (T + R), 4, D), dtype=bf16, device=dxs.device)
    _lib.get().e2k_stream_unpack_bwd(_p(dxs), _p(dX), B, T, R, D, _stream(dxs))
    return dX


def time_cond_fwd(times, fw, W, bias):
    """-> (out (B,D), four (B,D+1), pre (B,D)) fp32"""
    _chk(times, fw, W, bias)
    B, D = times.shape[0], W.shape[0]
    for t in (times, fw, W, bias):
        assert t.dtype == f32 and t.is_contiguous()
    assert W.shape == (D, D + 1) and fw.numel() == D // 2
    four = torch.empty((B, D + 1), dtype=f32, device=W.device)
    pre = torch.empty((B, D), dtype=f32, device=W.device)
    out = torch.empty((B, D), dtype=f32, device=W.device)
    _lib.get().e2k_time_cond_fwd(_p(times), _p(fw), _p(W), _p(bias), _p(four), _p(pre), _p(out), B, D, _stream(W))
    return out, four, pre


def time_cond_bwd(dout, four, pre, dW, dbias):
    _chk(dout, four, pre, dW, dbias)
    B, D = pre.shape
    assert dout.dtype == f32 and dout.is_contiguous() and dW.is_contiguous() and dbias.is_contiguous()
    _lib.get().e2k_time_cond_bwd(_p(dout), _p(four), _p(pre), _p(dW), _p(dbias), B, D, _stream(pre))


def cond_bwd_prep(dcond, gates, gbias, B, L, D):
    """-> dcb (B, 4LD) bf16, dct (4LD, KB) bf16 (KB = B rounded up to 8, zero padded); dcond's gate slots scaled in place"""
    _chk(dcond, gates, gbias)
    KB = (B + 7) // 8 * 8
    assert dcond.is_contiguous() and gates.is_contiguous() and gbias.is_contiguous()
    dcb = torch.empty((B, 4 * L * D), dtype=bf16, device=dcond.device)
    dct = torch.empty((4 * L * D, KB), dtype=bf16, device=dcond.device)
    _lib.get().e2k_cond_bwd_prep(_p(dcond), _p(gates), _p(dcb), _p(dct), _p(gbias), B, L, D, KB, _stream(dcond))
    return dcb, dct


def flow_pack(x0, x1, t, span_mask, cpad):
    """E2TTS.forward's prologue in one launch (e2k_flow_pack): -> (w bf16 (M, cpad), cond bf16 (M, cpad), flow fp32 (B, T, C), cond fp32 (B, T, C))"""
    _chk(x0, x1, t, span_mask)
    B, T, C = x1.shape
    assert x0.shape == x1.shape and x0.dtype == f32 and x1.dtype == f32 and x0.is_contiguous() and x1.is_contiguous()
    assert t.dtype == f32 and t.numel() == B and t.is_contiguous() and span_mask.numel() == B * T and span_mask.is_contiguous()
    m8 = span_mask.view(torch.uint8) if span_mask.dtype == torch.bool else span_mask
    wb = torch.empty((B * T, cpad), dtype=bf16, device=x1.device)
    cb = torch.empty((B * T, cpad), dtype=bf16, device=x1.device)
    flow, cond = torch.empty_like(x1), torch.empty_like(x1)
    _lib.get().e2k_flow_pack(_p(x0), _p(x1), _p(t), _p(m8), _p(wb), _p(cb), cpad, _p(flow), _p(cond), B, T, C, cpad, _stream(x1))
    return wb, cb, flow, cond


def cast_pad_bf16(src, cpad, out=None, col0=0):
    """src (R, C) fp32 (rows contiguous) -> bf16 (R, cpad) zero padded; with `out` (R, ld) the result goes to its columns
    col0 .. col0 + cpad"""
    _chk(src, out)
    R, C = src.shape
    assert src.dtype == f32 and src.stride(1) == 1 and cpad >= C
    if out is None:
        out = torch.empty((R, cpad), dtype=bf16, device=src.device)
    assert out.dtype == bf16 and out.stride(1) == 1 and out.shape[0] == R and col0 + cpad <= out.shape[1]
    _lib.get().e2k_cast_pad_bf16(_p(src), src.stride(0), out.data_ptr() + 2 * col0, out.stride(0), R, C, cpad, _stream(src))
    return out


def masked_mse_fwd(pred, flow, mask):
    """-> acc (2,) fp32 = [sum of squared errors over the masked rows, number of masked rows]"""
    _chk(pred, flow, mask)
    M, C = pred.shape
    assert pred.dtype == f32 and flow.dtype == f32 and pred.is_contiguous() and flow.is_contiguous() and mask.numel() == M and mask.is_contiguous()
    acc = torch.empty(2, dtype=f32, device=pred.device)
    _lib.get().e2k_masked_mse_fwd(_p(pred), _p(flow), _p(mask), _p(acc), M, C, _stream(pred))
    return acc


def masked_mse_bwd(pred, flow, mask, acc, dloss):
    _chk(pred, flow, mask, acc, dloss)
    M, C = pred.shape
    dpred = torch.empty_like(pred)
    _lib.get().e2k_masked_mse_bwd(_p(pred), _p(flow), _p(mask), _p(acc), _p(dloss), _p(dpred), M, C, _stream(pred))
    return dpred


def transpose_f32(src, C):
    """src (R, ld) fp32 -> (C, R) fp32 of its first C columns"""
    _chk(src)
    R = src.shape[0]
    assert src.dtype == f32 and src.stride(1) == 1
    out = torch.empty((C, R), dtype=f32, device=src.device)
    _lib.get().e2k_transpose_f32(_p(src), src.stride(0), _p(out), R, C, _stream(src))
    return out


# ------------------------------------------------------------------------------------------------ MelSpec

_twiddles = {}
_mel_bands = {}


def melspec(wave, window, fb, n_fft, hop, lens=None, pad_value=0.):
    """wave (B, nw) fp32 -> (B, n_mels, 1 + nw // hop) fp32 log-mel.  lens (B,) int: ragged batch -- row b holds lens[b]
    valid samples and is transformed as if it were alone; frames past 1 + lens[b] // hop are filled with pad_value"""
    _chk(wave, window, fb, lens)
    assert wave.dim() == 2
    wave = wave.float().contiguous()
    B, nw = wave.shape
    key = (str(wave.device), n_fft)
    if key not in _twiddles:
        ang = 2 * torch.pi * torch.arange(n_fft // 2, dtype=torch.float64) / n_fft
        _twiddles[key] = (ang.cos().float().to(wave.device), ang.sin().float().to(wave.device))
    twc, tws = _twiddles[key]
    n_mels = fb.shape[1]
    fb_in = fb
    fb = fb.float().contiguous()
    # the band table is cached per CALLER tensor (weak reference + version counter): a key made of the address of a converted
    # temporary could be met again by another filterbank that the allocator put at the same place
    ent = _mel_bands.get(id(fb_in))
    bands = None
    if ent is not None and ent[0]() is fb_in and ent[1] == fb_in._version and ent[2].device == fb.device:
        bands = ent[2]
    if bands is None:                        # non-zero bin range of every filterbank column (the htk triangles are narrow)
        nz = (fb != 0).cpu()
        ks = torch.arange(fb.shape[0])[:, None]
        lo = torch.where(nz, ks, fb.shape[0]).amin(0)
        hi = torch.where(nz, ks + 1, 0).amax(0)
        bands = torch.stack([lo, hi], 1).to(device=fb.device, dtype=torch.int32).contiguous()
        if len(_mel_bands) > 8:
            _mel_bands.clear()
        try:
            import weakref
            _mel_bands[id(fb_in)] = (weakref.ref(fb_in), fb_in._version, bands)
        except TypeError:
            pass
    out = torch.empty((B, n_mels, 1 + nw // hop), dtype=f32, device=wave.device)
    if lens is not None:
        assert lens.shape == (B,)
        lens32 = lens.to(device=wave.device, dtype=torch.int32).contiguous()
        _lib.get().e2k_melspec_ragged(_p(wave), nw, _p(lens32), _p(window.float().contiguous()), _p(fb),
                                      _p(twc), _p(tws), _p(out), float(pad_value), B, n_fft, hop, n_mels, _p(bands), _stream(wave))
        return out
    _lib.get().e2k_melspec(_p(wave), nw, _p(window.float().contiguous()), _p(fb), _p(twc), _p(tws),
                           _p(out), B, n_fft, hop, n_mels, _p(bands), _stream(wave))
    return out


# ------------------------------------------------------------------------------------------------ optimizer side (K19)

def sumsq(x, out):
    """out[0] += sum(x^2)   (x fp32 contiguous, out fp64 scalar tensor)"""
    _chk(x, out)
    assert x.dtype == f32 and x.is_contiguous() and out.dtype == torch.float64 and out.numel() == 1
    _lib.get().e2k_sumsq_f32(_p(x), x.numel(), _p(out), _stream(x))
    return out


def adopt_step(p, g, m, v, step, *, lr, beta1=0.9, beta2=0.99, eps=1e-6, weight_decay=0., max_grad_norm=0., gsumsq=None,
               shadow=None, ranges=None, step_b=0, active_b=True, ema=None, ema_decay=0.):
    """one fused ADOPT step on flat fp32 buffers (see e2k_adopt_step / e2k_adopt_step_groups / e2k_adopt_step_ema in include/e2k.h).
    ranges: int32 device tensor (nr, 2) of sorted [start, end) element ranges that form a second parameter group with its
    own step count `step_b`, skipped entirely when not active_b.  ema: fp32 buffer like p that follows the NEW parameters in the
    same pass, ema += (1 - ema_decay) (p - ema)"""
    _chk(p, g, m, v, gsumsq, shadow, ranges, ema)
    n = p.numel()
    for t in (p, g, m, v) + ((ema,) if ema is not None else ()):
        assert t.dtype == f32 and t.is_contiguous() and t.numel() == n
    if shadow is not None:
        assert shadow.dtype == bf16 and shadow.is_contiguous() and shadow.numel() == n
    if gsumsq is not None:
        assert gsumsq.dtype == torch.float64 and gsumsq.numel() == 1
    if ranges is not None and ranges.numel():
        assert ranges.dtype == torch.int32 and ranges.is_contiguous() and ranges.dim() == 2 and ranges.shape[1] == 2
    else:
        ranges = None
    if ema is not None:
        _lib.get().e2k_adopt_step_ema(_p(p), _p(g), _p(m), _p(v), _p(shadow), n, float(lr), float(beta1), float(beta2), float(eps),
                                      float(weight_decay), float(max_grad_norm), _p(gsumsq), int(step), int(step_b),
                                      int(bool(active_b)), _p(ranges), ranges.shape[0] if ranges is not None else 0, _p(ema),
                                      float(ema_decay), _stream(p))
        return
    if ranges is not None:
        _lib.get().e2k_adopt_step_groups(_p(p), _p(g), _p(m), _p(v), _p(shadow), n, float(lr), float(beta1), float(beta2), float(eps),
                                         float(weight_decay), float(max_grad_norm), _p(gsumsq), int(step), int(step_b),
                                         int(bool(active_b)), _p(ranges), ranges.shape[0], _stream(p))
        return
    _lib.get().e2k_adopt_step(_p(p), _p(g), _p(m), _p(v), _p(shadow), n, float(lr), float(beta1), float(beta2), float(eps),
                              float(weight_decay), float(max_grad_norm), _p(gsumsq), int(step), _stream(p))


def ema_update(ema, p, decay):
    _chk(ema, p)
    assert ema.dtype == f32 and p.dtype == f32 and ema.is_contiguous() and p.is_contiguous() and ema.numel() == p.numel()
    _lib.get().e2k_ema_update(_p(ema), _p(p), p.numel(), float(decay), _stream(p))


def grad_pack_bf16(g, wire, scale):
    """wire[i] = bf16(g[i] * scale): gradient slab -> wire format of the data-parallel exchange (one pass)"""
    _chk(g, wire)
    assert g.dtype == f32 and wire.dtype == bf16 and g.is_contiguous() and wire.is_contiguous() and wire.numel() >= g.numel()
    _lib.get().e2k_grad_pack_bf16(_p(g), _p(wire), g.numel(), float(scale), _stream(g))


def grad_unpack_bf16(wire, g):
    _chk(g, wire)
    assert g.dtype == f32 and wire.dtype == bf16 and g.is_contiguous() and wire.is_contiguous() and wire.numel() >= g.numel()
    _lib.get().e2k_grad_unpack_bf16(_p(wire), _p(g), g.numel(), _stream(g))


def resample_sinc(x, kernel, orig, new, width, lens=None):
    """x (B, n) fp32 -> (B, ceil(n new / orig)) fp32: polyphase windowed-sinc rate conversion (e2k_resample_sinc); kernel (new, 2 width +
    orig) fp32; lens (B,) int32: valid samples per row (outputs past a row's end are zero)"""
    _chk(x, kernel, lens)
    assert x.dim() == 2 and x.dtype == f32 and x.stride(1) == 1 and kernel.dtype == f32 and kernel.is_contiguous()
    assert kernel.shape == (new, 2 * width + orig)
    B, n = x.shape
    n_out = (n * new + orig - 1) // orig
    out = torch.empty((B, n_out), dtype=f32, device=x.device)
    if lens is not None:
        assert lens.dtype == torch.int32 and lens.numel() == B and lens.is_contiguous()
    _lib.get().e2k_resample_sinc(_p(x), x.stride(0), n, _p(lens), _p(kernel), _p(out), out.stride(0), n_out, B, int(orig), int(new),
                                 kernel.shape[1], int(width), _stream(x))
    return out


def shard_sum_bf16(recv, out, world):
    """out[i] = bf16(sum_r float(recv[r * per + i])), per = out.numel(): fp32 sum of the peers' copies of this rank's shard"""
    _chk(recv, out)
    per = out.numel()
    assert recv.dtype == bf16 and out.dtype == bf16 and recv.is_contiguous() and out.is_contiguous() and recv.numel() == per * world and per % 8 == 0
    _lib.get().e2k_shard_sum_bf16(_p(recv), _p(out), per, int(world), _stream(out))


def cfg_combine(pred, null_pred, cfg_strength, keep_parallel_frac=0., remove_parallel=True):
    """pred + cfg_update * cfg_strength with the fp64 parallel-component projection of e2_tts.py:113-124,1303-1330 in one
    kernel; pred / null_pred (B, ...) fp32 contiguous"""
    _chk(pred, null_pred)
    assert pred.dtype == f32 and null_pred.dtype == f32 and pred.is_contiguous() and null_pred.is_contiguous() and pred.shape == null_pred.shape
    out = torch.empty_like(pred)
    B = pred.shape[0]
    _lib.get().e2k_cfg_combine(_p(pred), _p(null_pred), _p(out), B, pred.numel() // B, float(cfg_strength), float(keep_parallel_frac),
                               int(bool(remove_parallel)), _stream(pred))
    return out


def fourier_cat_fwd(h, nf):
    """[sin h[:, :nf] | cos h[:, :nf] | h[:, nf:]]  (LinearFourierEmbed, e2_tts.py:383-386); h (M, nf + nrest) fp32 -> bf16"""
    _chk(h)
    M, nh = h.shape
    assert h.dtype == f32 and h.stride(1) == 1
    y = torch.empty((M, nh + nf), dtype=bf16, device=h.device)
    _lib.get().e2k_fourier_cat_fwd(_p(h), h.stride(0), _p(y), y.stride(0), M, nf, nh - nf, _stream(h))
    return y


def fourier_cat_bwd(dy, h, nf):
    _chk(dy, h)
    M, nh = h.shape
    assert dy.dtype == bf16 and h.dtype == f32 and dy.shape == (M, nh + nf) and dy.stride(1) == 1 and h.stride(1) == 1
    dh = torch.empty((M, nh), dtype=bf16, device=h.device)
    _lib.get().e2k_fourier_cat_bwd(_p(dy), dy.stride(0), _p(h), h.stride(0), _p(dh), dh.stride(0), M, nf, nh - nf, _stream(h))
    return dh


def freq_attn_fwd(qkv, B, F, N, H, cosb, sinb, vfirst=None):
    """attention over the F frequency tokens of every (batch row, frame) (e2_tts.py:920-932); qkv (B*F*N, 3*H*64) bf16 in
    token order (b f) n; vfirst = the first layer's v columns (a strided view) or None"""
    _chk(qkv, cosb, sinb, vfirst)
    I = H * 64
    assert qkv.dtype == bf16 and qkv.shape == (B * F * N, 3 * I) and qkv.stride(1) == 1
    assert cosb.shape == (F, 32) and cosb.is_contiguous() and sinb.is_contiguous()
    out = torch.empty((B * F * N, I), dtype=bf16, device=qkv.device)
    _lib.get().e2k_freq_attn_fwd(_p(qkv), qkv.stride(0), _p(vfirst), 0 if vfirst is None else vfirst.stride(0), _p(cosb), _p(sinb),
                                 _p(out), B, F, N, H, _stream(qkv))
    return out


def freq_attn_bwd(dout, qkv, B, F, N, H, cosb, sinb, vfirst=None, dvfirst=None, first_layer=False):
    _chk(dout, qkv, cosb, sinb, vfirst, dvfirst)
    I = H * 64
    assert dout.dtype == bf16 and dout.shape == (B * F * N, I) and dout.is_contiguous()
    assert dvfirst is None or (dvfirst.dtype == f32 and dvfirst.shape == (B * F * N, I) and dvfirst.is_contiguous())
    dqkv = torch.empty((B * F * N, 3 * I), dtype=bf16, device=qkv.device)
    _lib.get().e2k_freq_attn_bwd(_p(dout), _p(qkv), qkv.stride(0), _p(vfirst), 0 if vfirst is None else vfirst.stride(0), _p(cosb),
                                 _p(sinb), _p(dvfirst), int(first_layer), _p(dqkv), dqkv.stride(0), B, F, N, H, _stream(dout))
    return dqkv


def char_embed_fwd(tok, W, T):
    """CharacterEmbed (e2_tts.py:400-412): tok (B, nt) int64 (-1 = padding) -> (B, T, D) fp32 rows of W (V, D)"""
    _chk(tok, W)
    assert tok.dtype == torch.int64 and tok.is_contiguous() and W.dtype == f32 and W.is_contiguous()
    B, nt = tok.shape
    nt = min(nt, T)
    tk = tok if tok.shape[1] == nt else tok[:, :nt].contiguous()
    V, D = W.shape
    out = torch.empty((B, T, D), dtype=f32, device=W.device)
    _lib.get().e2k_char_embed_fwd(_p(tk), _p(W), _p(out), B, nt, T, D, V, _stream(W))
    return out, tk


def char_embed_bwd(tk, dout, V):
    _chk(tk, dout)
    assert dout.dtype == f32 and dout.is_contiguous()
    B, T, D = dout.shape
    dW = zeros((V, D), f32, dout.device)
    _lib.get().e2k_char_embed_bwd(_p(tk), _p(dout), _p(dW), B, tk.shape[1], T, D, V, _stream(dout))
    return dW


def duration_head_fwd(embed, mask8, w):
    """masked mean over frames + softplus(w . pooled) (e2_tts.py:1098-1111); embed (B, T, D) fp32, mask8 (B, T) uint8 or None"""
    _chk(embed, mask8, w)
    assert embed.dtype == f32 and embed.is_contiguous() and w.dtype == f32 and w.is_contiguous()
    B, T, D = embed.shape
    dev = embed.device
    pooled, z, pred = torch.empty((B, D), dtype=f32, device=dev), torch.empty(B, dtype=f32, device=dev), torch.empty(B, dtype=f32, device=dev)
    _lib.get().e2k_duration_head_fwd(_p(embed), _p(mask8), _p(w), _p(pooled), _p(z), _p(pred), B, T, D, _stream(embed))
    return pred, pooled, z


def duration_head_bwd(dpred, z, pooled, mask8, w, T):
    _chk(dpred, z, pooled, mask8, w)
    B, D = pooled.shape
    dev = pooled.device
    dembed = torch.empty((B, T, D), dtype=f32, device=dev)
    dw = zeros((D,), f32, dev)
    _lib.get().e2k_duration_head_bwd(_p(dpred), _p(z), _p(pooled), _p(mask8), _p(w), _p(dembed), _p(dw), B, T, D, _stream(pooled))
    return dembed, dw
