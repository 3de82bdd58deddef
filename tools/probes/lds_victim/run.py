"""victim kernels (LDS exchange / LDS table reads / DPP reduction) next to co-runners (transposing LDS reads, plain LDS reads,
LDS-DMA) on another stream: error counts"""
import ctypes, sys
from pathlib import Path
import torch
sys.path[:0] = [str(Path(__file__).resolve().parents[3] / 'e2-tts-pytorch_amd')]
from e2_tts_pytorch_amd import ops
here = Path(__file__).resolve().parent
L = ctypes.CDLL(str(here / 'liblds_victim.so'))
L.probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = 'cuda'
err = torch.zeros(1, dtype=torch.int32, device=dev)
sink = torch.zeros(4, device=dev)
src = torch.randint(0, 255, (1 << 21,), dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
victims = {0: 'exchange (masked ds_write_b128 + barrier + ds_read)', 1: 'table (ds_read_b128)', 2: 'dpp reduction'}
spam = {None: 'alone', 12: 'global_load_lds', 'nt_glds': 'real NT GEMM (glds)'}
an = torch.randn(2048, 1024, device=dev).to(torch.bfloat16); wn = torch.randn(2048, 1024, device=dev).to(torch.bfloat16); on = torch.empty(2048, 2048, device=dev, dtype=torch.bfloat16)
bf16 = torch.bfloat16
a1 = torch.randn(1024, 1552, device=dev).to(bf16); b1 = torch.randn(1024, 512, device=dev).to(bf16); a2 = a1[:928]; b2 = b1[:928]
out = torch.zeros(1552, 512, device=dev)
for v, vn in victims.items():
    for s, sn in spam.items():
        total = 0
        for trial in range(20):
            err.zero_()
            torch.cuda.synchronize()
            if isinstance(s, str):
                with torch.cuda.stream(side):
                    for _ in range(8):
                        ops.gemm_nt(an, wn, out=on)
            elif s is not None:
                for _ in range(3):
                    L.probe_launch(s, 512, 20000, sink.data_ptr(), src.data_ptr(), side.cuda_stream)
            L.probe_launch(v, 512, 300 if isinstance(s, str) else 2000, err.data_ptr(), None, main.cuda_stream)
            torch.cuda.synchronize()
            total += int(err.item())
        print(f'victim {vn:55s} next to {sn:22s}: {total} errors', flush=True)
