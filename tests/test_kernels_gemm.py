"""MFMA GEMM kernels against fp32 torch matmul on bf16-rounded operands (transpose-detecting: random asymmetric data)."""
import pytest
import torch

bf16 = torch.bfloat16


def rel(a, b):
    a, b = a.cpu().float(), b.cpu().float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


@pytest.mark.parametrize('M,N,K1,K2,kw', [
    (128, 128, 64, 0, {}),
    (200, 136, 72, 0, dict(bias=1)),
    (130, 260, 128, 64, dict(bias=1, cs=1, rm=1, rs=1)),
    (64, 100, 64, 0, dict(f32=1, bias=1)),
    (8, 1000, 256, 0, dict(f32=1, bias=1)),
    (256, 384, 128, 0, dict(f32=1, bias=1)),
    (520, 392, 256, 128, dict(rs=1)),
])
def test_gemm_nt(dev, M, N, K1, K2, kw):
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    a = torch.randn(M, K1).to(bf16)
    b = torch.randn(N, K1 + K2).to(bf16)
    a2 = torch.randn(M, K2).to(bf16) if K2 else None
    bias = torch.randn(N) if kw.get('bias') else None
    nb = 3
    rpb = (M + nb - 1) // nb
    cs = torch.rand(nb, N) if kw.get('cs') else None
    rm = (torch.rand(M) > 0.3) if kw.get('rm') else None
    rs = torch.randn(M, N).to(bf16) if kw.get('rs') else None
    to = lambda t: None if t is None else t.to(dev)
    out = ops.gemm_nt(to(a), to(b), a2=to(a2), bias=to(bias), colscale=to(cs), rows_per_batch=rpb, rowmask=to(rm),
                      resid=to(rs), out_dtype=torch.float32 if kw.get('f32') else bf16)
    A = torch.cat([a, a2], 1).float() if K2 else a.float()
    ref = A @ b.float().T
    if bias is not None:
        ref = ref + bias
    if cs is not None:
        ref = ref * cs[torch.arange(M) // rpb]
    if rm is not None:
        ref = ref * rm[:, None].float()
    if rs is not None:
        ref = ref + rs.float()
    assert rel(out, ref) < (1e-5 if kw.get('f32') else 6e-3)
    if kw.get('f32'):       # accumulate mode
        out2 = ops.gemm_nt(to(a), to(b), bias=to(bias), out=out.clone(), accumulate=True)
        assert rel(out2, 2 * ref) < 1e-5


@pytest.mark.parametrize('M,N,K1,K2,kw', [
    (1280, 128, 256, 0, dict(bias=1)),                       # 10 tiles on "8 slots": 2 remainder tiles x 2 K ranges
    (700, 260, 128, 128, dict(bias=1, cs=1, rm=1, rs=1)),    # 18 tiles: 2 remainder tiles (ragged edge), dual-K
    (1152, 128, 320, 0, dict(f32=1, bias=1)),                # 9 tiles: 1 remainder tile, 5 K steps -> uneven ranges
])
def test_gemm_nt_remainder_split(dev, M, N, K1, K2, kw):
    from e2_tts_pytorch_amd import ops
    old = ops.gemm_flags
    ops.gemm_flags = 32          # E2K_GEMM_TEST_SLOTS8
    try:
        test_gemm_nt(dev, M, N, K1, K2, kw)
    finally:
        ops.gemm_flags = old


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K1,K2,kw', [(8448, 3104, 1024, 0, dict(bias=1)), (8448, 4096, 512, 0, {}), (33792, 1024, 1024, 512, dict(rs=1))])
def test_gemm_nt_remainder_split_at_step_shapes(M, N, K1, K2, kw):
    """E2K_GEMM_SPLIT (512; off by default since round 6) on three shapes of a cfg3 step whose tile count leaves a partial round on the
    chip's 256 workgroup slots (429, 528 and 528 tiles): the real slot count, the fix-up kernel, the fp32 partials -- against the fp32
    product, and the default (unsplit) launch of the same shape"""
    from conftest import install_lib
    from e2_tts_pytorch_amd import ops
    install_lib(None, host_pointers=False)
    old = ops.gemm_flags
    try:
        for flags in (512, 0):
            ops.gemm_flags = flags
            test_gemm_nt('cuda', M, N, K1, K2, kw)
    finally:
        ops.gemm_flags = old


@pytest.mark.parametrize('use_tr', [False, True])
@pytest.mark.parametrize('M,N,K,splits', [(128, 128, 128, 1), (300, 136, 72, 0), (1000, 392, 264, 3), (8, 256, 128, 1),
                                          (256, 136, 72, 0), (1024, 392, 264, 3), (192, 8, 520, 1)])
def test_gemm_tn(dev, M, N, K, splits, use_tr):
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    a = torch.randn(M, N).to(bf16)
    b = torch.randn(M, K).to(bf16)
    out = torch.ones(N, K, device=dev)
    ops.gemm_tn(a.to(dev), b.to(dev), out, splits=splits, use_tr=use_tr)
    ref = 1 + a.float().T @ b.float()
    assert rel(out, ref) < 1e-5


@pytest.mark.parametrize('late', [0, 1])
@pytest.mark.parametrize('M,N,K,splits,cs_from', [
    (64, 256, 256, 1, None),        # one tile, ONE reduction step (prologue and drain only)
    (128, 256, 256, 1, 0),          # two steps; bias gradient riding along in every wave
    (192, 264, 520, 1, None),       # 2 x 3 tiles with ragged edges, odd number of steps
    (640, 520, 264, 2, 130),        # token split into partial tiles + reduce, column sums from a later column on
    (1024, 392, 264, 0, 8),         # split count chosen by the library
    (576, 136, 72, 3, None),        # narrower than one tile in both directions
])
def test_gemm_tn_256_tile(dev, monkeypatch, M, N, K, splits, cs_from, late):
    """256 x 256 x 64 8-phase weight-gradient kernel (use_tr = 3): same half-tile ring and counted waits as the NT one; on the
    host model under both LDS-DMA landing extremes"""
    from e2_tts_pytorch_amd import ops
    if dev == 'cuda' and late:
        pytest.skip('LDS-DMA landing extremes exist on the host model only')
    monkeypatch.setenv('E2K_EMU_GLDS_LATE', str(late))
    torch.manual_seed(0)
    a = torch.randn(M, N).to(bf16)
    b = torch.randn(M, K).to(bf16)
    out = torch.ones(N, K, device=dev)
    cs = torch.full((N,), 0.25, device=dev) if cs_from is not None else None
    ops.gemm_tn(a.to(dev), b.to(dev), out, splits=splits, use_tr=3, colsum=cs, colsum_from=cs_from or 0)
    ref = 1 + a.float().T @ b.float()
    assert rel(out, ref) < 1e-5
    if cs is not None:
        want = torch.full((N,), 0.25)
        want[cs_from:] += a.float().sum(0)[cs_from:]
        assert torch.allclose(cs.cpu(), want, rtol=1e-4, atol=1e-3), (cs.cpu() - want).abs().max()


def test_gemm_tn_strided(dev):
    """column-sliced operands / outputs as the backbone uses them (cross-condition and skip weight gradients)"""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    M, N, K, ldc = 256, 128, 64, 200
    a = torch.randn(M, N + 8).to(bf16)
    b = torch.randn(M, K + 16).to(bf16)
    C = torch.zeros(N, ldc, device=dev)
    ops.gemm_tn(a.to(dev)[:, :N], b.to(dev)[:, 8:8 + K], C[:, 40:40 + K])
    ref = a[:, :N].float().T @ b[:, 8:8 + K].float()
    assert rel(C[:, 40:40 + K], ref) < 1e-5
    assert float(C[:, :40].abs().max()) == 0 and float(C[:, 40 + K:].abs().max()) == 0


@pytest.mark.parametrize('M,N,K,cs_from', [(256, 264, 136, 0), (192, 392, 128, 130), (200, 136, 72, 8)])
def test_gemm_tn_colsum(dev, M, N, K, cs_from):
    """bias gradient riding along in the weight-gradient kernel (M % 64 == 0) / the separate pass of the general path"""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(1)
    a = torch.randn(M, N).to(bf16)
    b = torch.randn(M, K).to(bf16)
    out = torch.zeros(N, K)
    cs = torch.full((N,), 0.5)
    ops.gemm_tn(a.to(dev), b.to(dev), out.to(dev), colsum=(csd := cs.to(dev)), colsum_from=cs_from)
    want = torch.full((N,), 0.5)
    want[cs_from:] += a.float().sum(0)[cs_from:]
    assert torch.allclose(csd.cpu(), want, rtol=1e-4, atol=1e-3), (csd.cpu() - want).abs().max()



@pytest.mark.parametrize('flags', [128, 128 | 32])
@pytest.mark.parametrize('M,N,K1,K2,kw', [
    (256, 256, 64, 0, {}),                                   # one tile, ONE K tile (prologue and drain only)
    (300, 300, 128, 0, dict(bias=1)),                        # ragged edges in both directions, two K tiles
    (520, 260, 192, 64, dict(bias=1, cs=1, rm=1, rs=1)),     # dual-K (3 + 1 tiles), every epilogue operand
    (2304, 256, 512, 0, dict(f32=1, bias=1)),                # 9 tiles, 8 K tiles; with flag 32: 1 remainder tile x 2 K ranges
    (1280, 512, 256, 256, dict(bias=1, cs=1, rm=1, rs=1)),   # 10 tiles; with flag 32: 2 remainder tiles x 2 K ranges, dual-K
    (700, 520, 320, 0, dict(rs=1)),                          # odd number of K tiles (5)
])
@pytest.mark.parametrize('late', [0, 1])
def test_gemm_nt_256_tile(dev, monkeypatch, M, N, K1, K2, kw, flags, late):
    """256 x 256 tile, 8-phase NT kernel (flag E2K_GEMM_T256), alone and with the remainder split, on the host model and on
    the GPU.  What the host model checks: tile / quadrant / fragment indexing, the
    staging order against program-order overwrites, K-range tails, partials + fix-up, barrier pairing of the two wave
    groups (the model aborts on a mismatched barrier).  late = 1: LDS-DMA copies land as late as the issuing lane's
    counted `s_waitcnt vmcnt` allows instead of at issue (tests/emu/hip/hip_runtime.h) -- a wrong count reads stale data"""
    from e2_tts_pytorch_amd import ops
    if dev == 'cuda' and late:
        pytest.skip('LDS-DMA landing extremes exist on the host model only')
    monkeypatch.setenv('E2K_EMU_GLDS_LATE', str(late))
    old = ops.gemm_flags
    ops.gemm_flags = flags
    try:
        test_gemm_nt(dev, M, N, K1, K2, kw)
    finally:
        ops.gemm_flags = old


@pytest.mark.parametrize('M,N,K1,K2,kw', [
    (256, 256, 64, 0, {}),                                        # one whole tile, no epilogue operand
    (300, 296, 128, 0, dict(bias=1)),                             # ragged rows and columns (N = 296: chunks of 8 in or out as a whole)
    (520, 264, 192, 64, dict(bias=1, cs=1, rm=1, rs=1)),          # every epilogue operand, dual-K
    (2304, 256, 512, 0, dict(f32=1, bias=1)),                     # fp32 output
    (700, 520, 320, 0, dict(rs=1)),
    (256, 260, 64, 0, dict(bias=1)),                              # N not a multiple of 8: the direct epilogue must be chosen
])
def test_gemm_nt_256_staged_epilogue_matches_direct(dev, M, N, K1, K2, kw):
    """the 256 x 256 kernel writes its C tile through LDS in whole-line row segments (nt_epilogue_staged, the default
    where every epilogue operand allows 4-element accesses) -- same arithmetic in the same order as the direct epilogue
    (E2K_GEMM_NO_STAGE = 64): identical bits, also with accumulate into an fp32 C"""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(M + N)
    d = lambda t: None if t is None else t.to(dev)
    a = d(torch.randn(M, K1).to(bf16))
    a2 = d(torch.randn(M, K2).to(bf16)) if K2 else None
    b = d(torch.randn(N, K1 + K2).to(bf16))
    bias = d(torch.randn(N)) if kw.get('bias') else None
    nb = 3
    rpb = (M + nb - 1) // nb
    cs = d(torch.rand(nb, N)) if kw.get('cs') else None
    rm = d(torch.rand(M) > 0.3) if kw.get('rm') else None
    rs = d(torch.randn(M, N).to(bf16)) if kw.get('rs') else None
    f32o = bool(kw.get('f32'))
    old = ops.gemm_flags
    res = {}
    try:
        for flags in (128, 128 | 64):
            ops.gemm_flags = flags
            out = d(torch.full((M, N), 0.5)) if f32o else None
            o = ops.gemm_nt(a, b, a2=a2, bias=bias, colscale=cs, rows_per_batch=rpb, rowmask=rm, resid=rs, out=out,
                            accumulate=f32o, out_dtype=torch.float32 if f32o else bf16)
            res[flags] = o.cpu()
    finally:
        ops.gemm_flags = old
    assert torch.equal(res[128], res[128 | 64])
    A = torch.cat([a.cpu(), a2.cpu()], 1).float() if K2 else a.cpu().float()
    ref = A @ b.cpu().float().T
    if bias is not None:
        ref = ref + bias.cpu()
    if cs is not None:
        ref = ref * cs.cpu()[torch.arange(M) // rpb]
    if rm is not None:
        ref = ref * rm.cpu()[:, None].float()
    if rs is not None:
        ref = ref + rs.cpu().float()
    if f32o:
        ref = ref + 0.5
    assert rel(res[128], ref) < (1e-5 if f32o else 6e-3)


@pytest.mark.parametrize('late', [0, 1])
def test_gemm_nt_256_random_shapes(dev, monkeypatch, late):
    """seeded sweep of the 256 x 256 kernel over ragged M / N, 1..9 K tiles, dual-K splits at every tile boundary, every
    epilogue operand, with and without the remainder split (8-slot test hook); both LDS-DMA landing extremes"""
    import random as pyrandom
    from e2_tts_pytorch_amd import ops
    if dev == 'cuda' and late:
        pytest.skip('LDS-DMA landing extremes exist on the host model only')
    monkeypatch.setenv('E2K_EMU_GLDS_LATE', str(late))
    rng = pyrandom.Random(4321 + late)
    old = ops.gemm_flags
    try:
        for case in range(10):
            M = rng.choice([1, 100, 256, 257, 511, 700, 1100, 2304])
            N = rng.choice([8, 100, 256, 264, 500, 520])
            nk = rng.randint(1, 9)
            nk2 = rng.choice([0, 0, rng.randint(1, 4)])
            K1, K2 = 64 * nk, 64 * nk2
            torch.manual_seed(case)
            a = torch.randn(M, K1).to(bf16)
            a2 = torch.randn(M, K2).to(bf16) if K2 else None
            b = torch.randn(N, K1 + K2).to(bf16)
            bias = torch.randn(N) if rng.random() < 0.5 else None
            rs = torch.randn(M, N).to(bf16) if rng.random() < 0.4 else None
            rm = (torch.rand(M) > 0.3) if rng.random() < 0.3 else None
            f32out = rng.random() < 0.3
            ops.gemm_flags = 128 | rng.choice([0, 32])
            d = lambda t: None if t is None else t.to(dev)
            out = ops.gemm_nt(d(a), d(b), a2=d(a2), bias=d(bias), rowmask=d(rm), resid=d(rs),
                              out_dtype=torch.float32 if f32out else bf16).cpu()
            A = torch.cat([a, a2], 1).float() if K2 else a.float()
            ref = A @ b.float().T
            if bias is not None:
                ref = ref + bias
            if rm is not None:
                ref = ref * rm[:, None].float()
            if rs is not None:
                ref = ref + rs.float()
            assert rel(out, ref) < (1e-5 if f32out else 6e-3), (case, M, N, K1, K2, ops.gemm_flags)
    finally:
        ops.gemm_flags = old


def test_gemm_random_shapes(dev):
    """seeded sweep over ragged shapes / operand combinations of both GEMMs (edge tiles, odd K panels, every epilogue
    operand, the remainder split with the 8-slot test hook, column sums from an offset)"""
    import random as pyrandom
    from e2_tts_pytorch_amd import ops
    rng = pyrandom.Random(1234)
    to = lambda t: None if t is None else t.to(dev)
    old = ops.gemm_flags
    try:
        for case in range(24):
            M = rng.choice([1, 7, 64, 129, 200, 257, 520, 1100])
            N = rng.choice([8, 24, 100, 128, 136, 260, 392])
            K1 = 8 * rng.randint(1, 40)
            K2 = rng.choice([0, 0, 8 * rng.randint(1, 24)])
            torch.manual_seed(case)
            a = torch.randn(M, K1).to(bf16)
            a2 = torch.randn(M, K2).to(bf16) if K2 else None
            b = torch.randn(N, K1 + K2).to(bf16)
            bias = torch.randn(N) if rng.random() < 0.5 else None
            rs = torch.randn(M, N).to(bf16) if rng.random() < 0.4 else None
            rm = (torch.rand(M) > 0.3) if rng.random() < 0.3 else None
            f32out = rng.random() < 0.3
            ops.gemm_flags = rng.choice([0, 0, 32])
            out = ops.gemm_nt(to(a), to(b), a2=to(a2), bias=to(bias), rowmask=to(rm), resid=to(rs),
                              out_dtype=torch.float32 if f32out else bf16)
            A = torch.cat([a, a2], 1).float() if K2 else a.float()
            ref = A @ b.float().T
            if bias is not None:
                ref = ref + bias
            if rm is not None:
                ref = ref * rm[:, None].float()
            if rs is not None:
                ref = ref + rs.float()
            assert rel(out, ref) < (1e-5 if f32out else 6e-3), ('nt', case, M, N, K1, K2, ops.gemm_flags)
        ops.gemm_flags = 0
        for case in range(16):
            M = rng.choice([8, 64, 100, 192, 320, 1000])
            N = 8 * rng.randint(1, 50)
            K = 8 * rng.randint(1, 50)
            torch.manual_seed(100 + case)
            a = torch.randn(M, N).to(bf16)
            b = torch.randn(M, K).to(bf16)
            out = torch.ones(N, K, device=dev)
            cs_from = 2 * rng.randint(0, N // 2 - 1) if rng.random() < 0.5 else None
            cs = torch.zeros(N, device=dev) if cs_from is not None else None
            ops.gemm_tn(a.to(dev), b.to(dev), out, splits=rng.choice([0, 1, 2, 3]), use_tr=rng.random() < 0.8,
                        colsum=cs, colsum_from=cs_from or 0)
            assert rel(out, 1 + a.float().T @ b.float()) < 1e-5, ('tn', case, M, N, K)
            if cs is not None:
                want = a.float().sum(0)
                want[:cs_from] = 0
                assert torch.allclose(cs.cpu(), want, rtol=1e-4, atol=1e-3), ('colsum', case, M, N, K, cs_from)
    finally:
        ops.gemm_flags = old

@pytest.mark.parametrize('M,F,K,p,flags,bias,want_h', [
    (256, 128, 256, 0.0, 0, 1, 1),          # one tile: value rows | gate rows as the two B half tiles
    (300, 256, 320, 0.1, 0, 1, 1),          # ragged rows, two column tiles, odd number of K tiles, dropout
    (2304, 128, 512, 0.1, 32, 1, 1),        # 9 tiles on the 8-slot test hook: 1 remainder tile x 2 K ranges + GEGLU fix-up
    (1280, 256, 512, 0.0, 32, 0, 0),        # 10 tiles, 2 remainder tiles; no bias; inference form (H not stored)
])
@pytest.mark.parametrize('late', [0, 1])
def test_gemm_nt_geglu_epilogue(dev, monkeypatch, M, F, K, p, flags, bias, want_h, late):
    """FeedForward GEMM1 with the GEGLU (+ dropout) as its epilogue (e2_tts.py:646,692) against the two
    launches it replaces (the epilogue rounds H to bf16 before the product, as the separate kernel reads it) and against
    the fp32 formula with the oracle's dropout mask"""
    import torch.nn.functional as Fn
    from e2_tts_pytorch_amd import ops
    from oracle.dropout_hash import geglu_dropout_mask
    if dev == 'cuda' and late:
        pytest.skip('LDS-DMA landing extremes exist on the host model only')
    monkeypatch.setenv('E2K_EMU_GLDS_LATE', str(late))
    assert ops.can_fuse_geglu(M, F, K) and not ops.can_fuse_geglu(M, F + 64, K) and not ops.can_fuse_geglu(M, F, 96)
    torch.manual_seed(M + F)
    a = (torch.randn(M, K) * 0.5).to(bf16)
    w1 = (torch.randn(2 * F, K) * 0.1).to(bf16)
    b1 = torch.randn(2 * F) if bias else None
    seed, sid = 4242, 7
    d = lambda t: None if t is None else t.to(dev)
    old = ops.gemm_flags
    ops.gemm_flags = flags
    try:
        H, act = ops.gemm_nt_geglu(d(a), d(w1), d(b1), p, seed, sid, want_h=bool(want_h))
        ops.gemm_flags = flags | 128       # the same tile and K order for the unfused pair
        H2 = ops.gemm_nt(d(a), d(w1), bias=d(b1))
        act2 = ops.geglu_fwd(H2, p, seed, sid)
    finally:
        ops.gemm_flags = old
    # the activation is formed from the bf16-rounded H, exactly as the separate kernel reads it
    # the epilogue's erf is a 1.5e-7 approximation (csrc/gemm.hip: gelu_erf_fast): below bf16 rounding but for the odd last place,
    # and it differs (towards the exact value) for gates below about -3.7, where the separate kernel's fp32 `1 + erf` cancels
    same = lambda x, y: (x.cpu() != y.cpu()).float().mean().item() < 1e-2 and rel(x, y) < 8e-3
    if want_h:
        assert same(act, ops.geglu_fwd(H, p, seed, sid))
    else:
        assert H is None
    # against the unfused pair: same bits where both sum K in one pass; with the remainder split a fused tile (128 value +
    # 128 gate columns) and a plain tile (256 adjacent columns) cover different outputs, so a few elements are summed in
    # two K ranges by one and in one pass by the other (fp32 order -> the bf16 rounding of about 1 in 10^4 flips)
    if not (flags & 32):
        assert same(act, act2) and (H is None or torch.equal(H.cpu(), H2.cpu()))
    else:
        assert same(act, act2) and (H is None or ((H.cpu() != H2.cpu()).float().mean().item() < 1e-3 and rel(H, H2) < 1e-2))
    h = a.float() @ w1.float().T + (b1 if bias else 0.)
    ref = h[:, :F] * Fn.gelu(h[:, F:]) * (geglu_dropout_mask(seed, sid, M, F, p) if p else 1.)
    assert rel(act, ref) < 1e-2
    with pytest.raises(Exception):
        ops.gemm_nt_geglu(d(a), d(torch.zeros(2 * (F + 8), K).to(bf16)), None)


@pytest.mark.parametrize('M,F,K,p,flags', [
    (256, 256, 256, 0.0, 0),                # one tile
    (200, 256, 256, 0.25, 0),               # partial row tile, dropout
    (1280, 512, 256, 0.1, 32),              # 10 tiles on 8 test slots: remainder split + fix-up kernel
    (512, 256, 320, 0.0, 64),               # direct (un-staged) epilogue
    (66, 1024, 256, 0.0, 0),                # a short batch: one partial row tile, four column tiles
    (130, 512, 256, 0.1, 0),
])
@pytest.mark.parametrize('late', [0, 1])
def test_gemm_nt_geglu_bwd_epilogue(dev, monkeypatch, M, F, K, p, flags, late):
    """FeedForward's second dgrad GEMM with the GEGLU backward (+ dropout) as its epilogue (e2_tts.py:646,692,937) against the two
    launches it replaces and against the fp32 formula with the oracle's dropout mask.  The epilogue works on the fp32 accumulator
    where the separate kernel reads d(act) rounded to bf16: agreement to bf16 rounding, not bit for bit."""
    from e2_tts_pytorch_amd import ops
    from oracle.dropout_hash import geglu_dropout_mask
    if dev == 'cuda' and late:
        pytest.skip('LDS-DMA landing extremes exist on the host model only')
    monkeypatch.setenv('E2K_EMU_GLDS_LATE', str(late))
    q = ops.lib().e2k_query_gemm_nt_geglu_bwd
    assert q(M, F, K) in (1, 2) and q(M, F + 128, K) == 0 and q(M, F, 96) == 0           # 2: runs, but too few tiles to be recommended
    assert q(8448, 4096, 1024) == 1 and ops.can_fuse_geglu_bwd(8448, 4096, 1024) and not ops.can_fuse_geglu_bwd(300, 256, 256)
    torch.manual_seed(M + F + K)
    dy = (torch.randn(M, K) * 0.5).to(bf16)
    w2T = (torch.randn(F, K) * 0.1).to(bf16)
    H = torch.randn(M, 2 * F).to(bf16)
    seed, sid = 77, 9
    d = lambda t: t.to(dev)
    old = ops.gemm_flags
    ops.gemm_flags = flags
    try:
        dH = ops.gemm_nt_geglu_bwd(d(dy), d(w2T), d(H), p, seed, sid)
        ops.gemm_flags = flags | 128
        dH2 = ops.geglu_bwd(ops.gemm_nt(d(dy), d(w2T)), d(H), p, seed, sid)
    finally:
        ops.gemm_flags = old
    assert dH.shape == (M, 2 * F) and rel(dH, dH2) < 1.5e-2, rel(dH, dH2)
    dact = dy.double() @ w2T.double().T
    u, g = H[:, :F].double(), H[:, F:].double()
    keep = geglu_dropout_mask(seed, sid, M, F, p).double() if p else 1.
    Phi = 0.5 * (1 + torch.erf(g / 2 ** 0.5))
    ref = torch.cat([dact * keep * g * Phi, dact * keep * u * (Phi + g * torch.exp(-g * g / 2) / (2 * torch.pi) ** 0.5)], 1)
    assert rel(dH, ref.float()) < 1e-2, rel(dH, ref.float())
    # ... and closer to the fp64 formula than the two-launch pair is (no bf16 rounding of d(act) in between)
    assert (dH.cpu().float() - ref.float()).norm() <= 1.02 * (dH2.cpu().float() - ref.float()).norm()
    with pytest.raises(Exception):
        ops.gemm_nt_geglu_bwd(d(dy), d(torch.zeros(F + 8, K).to(bf16)), d(H))


def test_gelu_erf_fast_accuracy(dev):
    """the GEGLU epilogue's x * Phi(x) (Abramowitz & Stegun 7.1.26 with one rcp + one exp2; csrc/gemm.hip) against
    float64 erf over the range of bf16 gates: within one bf16 last place of the exact value from -5 up, and no
    `1 + erf` cancellation on the negative side (the tail keeps a relative accuracy of a few percent down to -12).  The gates go in through the
    bias (A = W1 = 0; value bias 1, gate bias g), one launch per row of 128 gates."""
    import math
    from e2_tts_pytorch_amd import ops
    M, F, K = 16, 128, 64
    gates = torch.cat([torch.linspace(-12., 12., 3072), torch.randn(1024) * 3.]).to(bf16).float().view(-1, F)
    a, w1 = torch.zeros(M, K).to(bf16).to(dev), torch.zeros(2 * F, K).to(bf16).to(dev)
    worst_tail = 0.
    for gb in gates:
        _, act = ops.gemm_nt_geglu(a, w1, torch.cat([torch.ones(F), gb]).to(dev), want_h=False)
        x = gb.double()
        ref = x * 0.5 * torch.erfc(-x / math.sqrt(2.))
        got = act[0].cpu().double()
        body = x >= -5.
        assert ((got - ref).abs() <= ref.abs() * 2. ** -7 + 1e-38)[body].all(), ((got - ref).abs() / ref.abs().clamp_min(1e-38))[body].max()
        tail = ~body & (ref.abs() > 1e-36)
        if tail.any():
            worst_tail = max(worst_tail, ((got - ref) / ref).abs()[tail].max().item())
    # beyond -5 (|value| < 1.5e-6 |x|) the formula's relative error grows slowly (3 % at -12, where the value is 1e-32);
    # the separate kernel's fp32 `1 + erf` is exactly 0 from about -5.5 on
    assert 0. < worst_tail < 0.05, worst_tail


@pytest.mark.parametrize('M', [1, 17, 257])
def test_gemm_nt_geglu_edge_shapes(dev, M):
    """GEGLU-epilogue GEMM on ragged row counts (1 row, a partial 16-row group, one row past a tile) with operands that
    are column slices of wider buffers (row strides larger than K), against the two separate launches"""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(M)
    F, K = 128, 128
    abuf = (torch.randn(M, K + 64) * 0.5).to(bf16).to(dev)
    wbuf = (torch.randn(2 * F, K + 192) * 0.1).to(bf16).to(dev)
    a, w1 = abuf[:, 64:], wbuf[:, 128:128 + K]              # 16-byte aligned starts, strides K + 64 / K + 192
    b1 = torch.randn(2 * F).to(dev)
    H, act = ops.gemm_nt_geglu(a, w1, b1, 0.2, 99, 3)
    old = ops.gemm_flags
    ops.gemm_flags = 128
    try:
        H2 = ops.gemm_nt(a, w1, bias=b1)
    finally:
        ops.gemm_flags = old
    act2 = ops.geglu_fwd(H2, 0.2, 99, 3)
    assert torch.equal(H.cpu(), H2.cpu())
    assert (act.cpu() != act2.cpu()).float().mean().item() < 2e-2 and rel(act, act2) < 8e-3
    assert H.shape == (M, 2 * F) and act.shape == (M, F)


@pytest.mark.parametrize('M,N1,N2,K1,K2,splits', [
    (256, 256, 0, 256, 0, 0),            # single source through the dual entry point
    (512, 256, 128, 256, 264, 2),        # both operands split; ragged second blocks; two token splits
    (320, 512, 0, 256, 256, 0),          # the skip projection's form: one dY, cat(x, skip)
    (640, 256, 256, 512, 0, 3),          # A split only
])
def test_gemm_tn_dual_source(dev, M, N1, N2, K1, K2, splits):
    """C[N1+N2, K1+K2] += cat(A1, A2)^T cat(B1, B2) in one launch of the 256 x 256 weight-gradient kernel
    (e2k_gemm_tn_dual_bf16: the cross-condition's four gradient blocks / the skip projection's two, neither concatenation
    materialised) against the block-by-block single-source GEMMs and the fp32 product, accumulated onto a non-zero C"""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(M + N1 + K2)
    d = lambda t: None if t is None else t.to(dev)
    a1, b1 = torch.randn(M, N1).to(bf16), torch.randn(M, K1).to(bf16)
    a2 = torch.randn(M, N2).to(bf16) if N2 else None
    b2 = torch.randn(M, K2).to(bf16) if K2 else None
    c0 = torch.randn(N1 + N2, K1 + K2)
    assert ops.can_gemm_tn_dual(M, N1, K1)
    out = d(c0.clone())
    ops.gemm_tn_dual(d(a1), d(a2), d(b1), d(b2), out, splits=splits)
    A = torch.cat([a1] + ([a2] if N2 else []), 1).float()
    Bm = torch.cat([b1] + ([b2] if K2 else []), 1).float()
    ref = c0 + A.T @ Bm
    assert rel(out, ref) < 2e-3
    blk = d(c0.clone())
    for a, r0 in ((a1, 0), (a2, N1)):
        for b_, k0 in ((b1, 0), (b2, K1)):
            if a is not None and b_ is not None:
                ops.gemm_tn(d(a), d(b_), blk[r0:r0 + a.shape[1], k0:k0 + b_.shape[1]], use_tr=3)
    assert rel(out, blk) < 1e-5


def test_gemm_tn_group(dev):
    """up to 8 weight gradients with the same token count in ONE launch of the 256 x 256 kernel (e2k_gemm_tn_group_bf16: a
    layer's attention-out / qkv / feed-forward gradients of both streams share the chip) against one GEMM per problem:
    ragged shapes, bias-gradient column sums on two of them, accumulated onto non-zero C, explicit and library-chosen
    split counts"""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(3)
    M = 640
    shapes = [(256, 512, 1, 0), (264, 136, 0, 0), (1040, 256, 1, 16), (136, 264, 0, 0), (520, 520, 0, 0)]
    for splits in (0, 1, 3):
        probs, refs = [], []
        for (N, K, cs, cs_from) in shapes:
            a, b = torch.randn(M, N).to(bf16), torch.randn(M, K).to(bf16)
            c0 = torch.randn(N, K)
            col0 = torch.full((N,), 0.25) if cs else None
            probs.append((a.to(dev), b.to(dev), c0.clone().to(dev), None if col0 is None else col0.clone().to(dev), cs_from))
            want_cs = None
            if cs:
                want_cs = col0.clone()
                want_cs[cs_from:] += a.float().sum(0)[cs_from:]
            refs.append((c0 + a.float().T @ b.float(), want_cs))
        assert all(ops.can_group_tn(p[0], p[1]) for p in probs)
        ops.gemm_tn_group(probs, splits=splits)
        for (a, b, out, col, _), (want, want_cs) in zip(probs, refs):
            assert rel(out, want) < 2e-3, splits
            if want_cs is not None:
                assert torch.allclose(col.cpu(), want_cs, rtol=1e-4, atol=2e-3), splits


@pytest.mark.parametrize('M,N1,N2,K1,K2,flags', [(300, 256, 128, 128, 64, 0), (520, 512, 256, 256, 128, 128), (264, 256, 256, 192, 0, 0),
                                               (700, 512, 256, 512, 256, 128 + 32), (600, 512, 384, 512, 0, 256 + 32)])     # (+ 32: remainder split + fix-up kernels)
def test_gemm_nt_two_outputs(dev, M, N1, N2, K1, K2, flags):
    """e2k_gemm_nt2_bf16 (TextAudioCrossCondition's two projections in one launch): bit-identical to two e2k_gemm_nt_bf16
    launches over the two row blocks of the weight, with and without residuals, on the 128 x 128 and the 256 x 256 kernel"""
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(M + N1)
    a = torch.randn(M, K1).to(bf16).to(dev)
    a2 = torch.randn(M, K2).to(bf16).to(dev) if K2 else None
    w = (torch.randn(N1 + N2, K1 + K2) * 0.1).to(bf16).to(dev)
    r1, r2 = torch.randn(M, N1).to(bf16).to(dev), torch.randn(M, N2).to(bf16).to(dev)
    old = ops.gemm_flags
    ops.gemm_flags = flags
    try:
        for res in (False, True):
            o1, o2 = ops.gemm_nt2(a, w, N1, a2=a2, resid=r1 if res else None, resid2=r2 if res else None)
            e1 = ops.gemm_nt(a, w[:N1], a2=a2, resid=r1 if res else None)
            e2 = ops.gemm_nt(a, w[N1:], a2=a2, resid=r2 if res else None)
            if flags & 32:          # a remainder tile is split over K in one form and not in the other: another fp32 summation order
                assert rel(o1.float().cpu(), e1.float().cpu()) < 1e-2 and rel(o2.float().cpu(), e2.float().cpu()) < 1e-2, res
            else:
                assert torch.equal(o1.float().cpu(), e1.float().cpu()) and torch.equal(o2.float().cpu(), e2.float().cpu()), res
        ref = torch.cat([a.float().cpu(), a2.float().cpu()], 1) @ w.float().cpu().T if K2 else a.float().cpu() @ w.float().cpu().T
        assert rel(e1.float().cpu(), ref[:, :N1] + r1.float().cpu()) < 2e-2
    finally:
        ops.gemm_flags = old


@pytest.mark.parametrize('B,N,H,K,vres,bias,need_v', [(2, 150, 4, 256, True, False, True), (1, 300, 4, 320, False, True, True),
                                                      (30, 10, 4, 256, True, True, False), (2, 257, 8, 256, False, False, True)])
@pytest.mark.parametrize('late', [0, 1])
def test_qkv_projection_with_the_rotary_epilogue(dev, monkeypatch, B, N, H, K, vres, bias, need_v, late):
    """e2k_gemm_nt_qkrot_bf16 + the value-only e2k_qkv_post_fwd against e2k_gemm_nt_bf16 + the full e2k_qkv_post_fwd: the same bits in
    everything the attention kernels and the backward pass read (x_transformers.Attention's projection + rotary, e2_tts.py:875,911).
    Rows of several batch elements inside one 16-row pass of the epilogue (N = 10), a ragged last row tile, an odd q | k boundary inside a
    tile are all in the shapes; against the fp32 rotation of the bf16-rounded projection as well"""
    if dev == 'cuda' and late:
        pytest.skip('LDS-DMA landing extremes exist on the host model only')
    monkeypatch.setenv('E2K_EMU_GLDS_LATE', str(late))
    from e2_tts_pytorch_amd import ops
    torch.manual_seed(0)
    I = H * 64
    cols = 3 * I + 2 * H
    M = B * N
    x = torch.randn(M, K).to(bf16).to(dev)
    w = (torch.randn(cols, K) / K ** 0.5).to(bf16).to(dev)
    bs = torch.randn(cols).to(dev) if bias else None
    cosb, sinb = ops.rotary_table(N, dev)
    vfirst = torch.randn(B, H, N, 64).to(bf16).to(dev) if vres else None
    assert ops._lib.get().e2k_query_gemm_nt_qkrot(M, cols, K, H) in (1, 2)
    ldq = (cols + 7) // 8 * 8 + 8
    # two launches
    qa = torch.zeros(M, ldq, dtype=bf16, device=dev)[:, :cols]
    ops.gemm_nt(x, w, bias=bs, out=qa)
    sa = ops.qkv_post_fwd(qa, B, H, N, cosb, sinb, vfirst, need_v=need_v)
    # fused
    qb = torch.full((M, ldq), 7., dtype=bf16, device=dev)[:, :cols]
    Q, Kh = ops.gemm_nt_qkrot(x, w, qb, B, H, N, cosb, sinb, bias=bs)
    sb = ops.qkv_post_fwd(qb, B, H, N, cosb, sinb, vfirst, need_v=need_v, qk=(Q, Kh))
    assert torch.equal(qb[:, 2 * I:], qa[:, 2 * I:])
    assert (qb[:, :2 * I] == 7).all()                    # the q | k columns of the row-major output are not written
    for name in ('Q', 'K', 'V', 'VT', 'gate', 'mix'):
        ta, tb = getattr(sa, name), getattr(sb, name)
        assert (ta is None) == (tb is None), name
        if ta is not None:
            assert torch.equal(ta, tb), name
    # and against the definition: rotation of interleaved pairs by the table, on the bf16-rounded projection
    ref = qa[:, :2 * I].float().cpu().view(B, N, 2, H, 32, 2)
    c, s = cosb.cpu().view(1, N, 1, 1, 32), sinb.cpu().view(1, N, 1, 1, 32)
    rot = torch.stack((ref[..., 0] * c - ref[..., 1] * s, ref[..., 1] * c + ref[..., 0] * s), -1).reshape(B, N, 2, H, 64)
    assert rel(Q, rot[:, :, 0].permute(0, 2, 1, 3)) < 8e-3 and rel(Kh, rot[:, :, 1].permute(0, 2, 1, 3)) < 8e-3


def test_qkv_projection_with_the_rotary_epilogue_refusals(dev):
    from e2_tts_pytorch_amd import ops
    q = ops._lib.get().e2k_query_gemm_nt_qkrot
    assert q(8448, 3104, 1024, 16) == 1 and q(8448, 1552, 512, 8) == 1       # cfg3: speech and text branches
    assert q(300, 776, 256, 4) == 2                                           # can run, not recommended (few tiles)
    assert q(8448, 3104, 1024, 15) == 0 and q(8448, 3104, 192, 16) == 0 and q(8448, 3104, 1000, 16) == 0 and q(8448, 3000, 1024, 16) == 0
    assert not ops.can_fuse_qk_rot(300, 776, 256, 4, 320) and ops.can_fuse_qk_rot(8448, 3104, 1024, 16, 1088)
    assert not ops.can_fuse_qk_rot(8 * 4160, 3104, 1024, 16, 4160)            # rows beyond 4096: the backward wants Q^T / K^T
    x = torch.zeros(300, 256, dtype=bf16, device=dev)
    w = torch.zeros(776, 192, dtype=bf16, device=dev)
    with pytest.raises(Exception):
        ops.gemm_nt_qkrot(x[:, :192], w, torch.zeros(300, 776, dtype=bf16, device=dev), 2, 4, 150, *ops.rotary_table(150, dev))
