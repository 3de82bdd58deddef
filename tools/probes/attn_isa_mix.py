"""Static instruction mix of the attention kernels' main loops (no GPU needed): compiles csrc/attn.hip to gfx950 assembly and
counts, per basic block of the key-tile loop, MFMA / quarter-rate (32-bit integer multiply, transcendental) / packed-fp32 /
other vector-ALU / LDS / scalar instructions.  Why: the PMC run of round 2 said "vector-ALU pipe 63 % busy"; this says which
instructions.  Writes profiles/r03_attn_isa_mix.json.

    python tools/probes/attn_isa_mix.py
"""
import collections
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CSRC = ROOT / 'e2-tts-pytorch_amd' / 'csrc'
KERNELS = {
    'attn_fwd_ring_kernel<DROP, SHARE, 2>': '_ZN12_GLOBAL__N_120attn_fwd_ring_kernelILb1ELb1ELi2E',
    'attn_bwd_dq_ring_kernel<DROP, SHARE>': '_ZN12_GLOBAL__N_123attn_bwd_dq_ring_kernelILb1ELb1E',
    'attn_bwd_dkv_ring_kernel<DROP, SHARE>': '_ZN12_GLOBAL__N_124attn_bwd_dkv_ring_kernelILb1ELb1E',
}
# blocks of the rarely taken paths (exp2 / rcp soft-clamp when a tile's logits leave the polynomial's range; key-mask selects
# of the last tile): excluded from the hot-path sums
COLD_HINT = ('v_rcp_f32',)


def cls(i):
    if i.startswith('v_mfma'):
        return 'mfma'
    if i.startswith(('v_mul_lo', 'v_mul_hi', 'v_mad_u64', 'v_mad_i64')):
        return 'int32_mul_quarter_rate'
    if i.startswith(('v_exp', 'v_rcp', 'v_log', 'v_sqrt', 'v_rsq')):
        return 'transcendental_quarter_rate'
    if i.startswith('v_pk_'):
        return 'packed_fp32'
    if i.startswith('v_cmp'):
        return 'v_cmp'
    if i.startswith('v_cndmask'):
        return 'v_cndmask'
    if i.startswith('v_mov'):
        return 'v_mov'
    if i.startswith('v_'):
        return 'other_valu'
    if i.startswith('ds_'):
        return 'lds'
    if i.startswith('s_'):
        return 'scalar'
    return 'other'


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = Path(td) / 'attn.s'
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', f'-I{CSRC}',
                        '--cuda-device-only', '-S', '-o', str(asm), str(CSRC / 'attn.hip')], check=True, stderr=subprocess.DEVNULL)
        lines = asm.read_text().split('\n')
    out = {}
    for title, sym in KERNELS.items():
        start = next(i for i, l in enumerate(lines) if l.startswith(sym) and ':' in l)
        end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith('s_endpgm'))
        blocks, cur, name = [], [], 'entry'
        for l in lines[start:end + 1]:
            t = l.strip()
            if re.match(r'^\.LBB\d+_\d+:', t):
                blocks.append((name, cur))
                cur, name = [], t
            elif t and not t.startswith(('.', ';')):
                cur.append(t.split()[0])
        blocks.append((name, cur))
        hot, cold = collections.Counter(), collections.Counter()
        for nm, b in blocks:
            if 'in Loop' not in nm:
                continue
            c = collections.Counter(cls(i) for i in b)
            is_cold = any(i.startswith(COLD_HINT) for i in b) and not any(i.startswith('v_mfma') for i in b)
            (cold if is_cold else hot).update(c)
        # blocks that hold both MFMAs and the exp2 / rcp fallback: the fallback's share is removed by hand below (fwd only)
        out[title] = dict(loop_hot_path=dict(hot), loop_rare_paths=dict(cold))
        v = sum(n for k, n in hot.items() if k not in ('mfma', 'lds', 'scalar', 'other'))
        q = hot.get('int32_mul_quarter_rate', 0) + hot.get('transcendental_quarter_rate', 0)
        out[title]['vector_alu_issue_clocks_per_tile'] = 4 * (v - q) + 16 * q
        out[title]['mfma_clocks_per_tile'] = 16 * hot.get('mfma', 0)
    out['note'] = ('per 64-key tile per wave (16 query rows); a wave64 vector instruction issues over 4 clocks on a 16-lane SIMD, the '
                   'quarter-rate ones over 16; a 16x16x32 bf16 MFMA is 16 clocks of its SIMD.  Blocks that mix MFMAs with the exp2 / rcp '
                   'soft-clamp fallback count the fallback too (upper bound).')
    (ROOT / 'profiles' / 'r03_attn_isa_mix.json').write_text(json.dumps(out, indent=1) + '\n')
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
