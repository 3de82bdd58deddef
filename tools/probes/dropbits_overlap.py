"""Does a register-only vector-ALU kernel (the dropout keep-bit generator, e2k_attn_dropbits) run for free next to a GEMM?
Times, at the cfg3 attention shape: the forward publishing its own bits (today), the generator, the forward reading ready bits;
then a GEMM loop alone, the generator loop alone and both on two streams.  -> gpurun_out/dropbits_overlap.json"""
import json, os, sys, time
from pathlib import Path
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / 'e2-tts-pytorch_amd'), str(ROOT)]
import torch
from e2_tts_pytorch_amd import ops
dev = torch.device('cuda')
torch.manual_seed(0)
B, H, N = 8, 16, 1056
I = H * 64
cols = 3 * I + 2 * H
qkvg = (torch.randn(B * N, cols, device=dev) * 0.5).to(torch.bfloat16)
cosb, sinb = ops.rotary_table(N, dev)
st = ops.qkv_post_fwd(qkvg, B, H, N, cosb, sinb, None)
Npad = (N + 63) // 64 * 64
kmask = torch.zeros(B, Npad, dtype=torch.uint8, device=dev); kmask[:, :N] = 1
p, seed, sid = 0.1, 1234, 8


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {}
res['fwd_publish_ms'] = timeit(lambda: ops.attn_fwd(st, kmask, p, seed, sid))
og_pub = ops.attn_fwd(st, kmask, p, seed, sid).clone(); bits_pub = st.dropbits.clone()
res['generator_ms'] = timeit(lambda: ops.attn_dropbits(B, H, N, p, seed, sid, None, dev))
bits = ops.attn_dropbits(B, H, N, p, seed, sid, None, dev)
res['bits_identical'] = bool(torch.equal(bits, bits_pub))
res['fwd_consume_ms'] = timeit(lambda: ops.attn_fwd(st, kmask, p, seed, sid, dropbits=bits))
res['fwd_consume_identical'] = bool(torch.equal(ops.attn_fwd(st, kmask, p, seed, sid, dropbits=bits), og_pub))
res['fwd_no_dropout_ms'] = timeit(lambda: ops.attn_fwd(st, kmask, 0., seed, sid))
# overlap with GEMMs of the step
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name, (M, Nn, K) in {'stream_gemm_33792x1024x1024': (33792, 1024, 1024), 'ff1_8448x8192x1024': (8448, 8192, 1024),
                         'qkv_8448x3104x1024': (8448, 3104, 1024)}.items():
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(Nn, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    n = 40

    def gemm_loop():
        with torch.cuda.stream(sa):
            for _ in range(n):
                ops.gemm_nt(a, w, out=out)

    def gen_loop():
        with torch.cuda.stream(sb):
            for _ in range(n):
                ops.attn_dropbits(B, H, N, p, seed, sid, None, dev)

    def wall(*fns):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in fns:
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    gemm_loop(); gen_loop(); torch.cuda.synchronize()
    g, r, both = wall(gemm_loop), wall(gen_loop), wall(gemm_loop, gen_loop)
    res[name] = dict(gemm_alone_ms=g, generator_alone_ms=r, both_streams_ms=both, hidden_fraction_of_generator=(g + r - both) / r)
print(json.dumps(res, indent=1))
(ROOT / 'gpurun_out').mkdir(exist_ok=True)
json.dump(res, open(ROOT / 'gpurun_out' / 'dropbits_overlap.json', 'w'), indent=1)
