"""Variant libraries for profiles/r06t_rotary_form_vs_lanes.txt: today's attn.hip with the rotary pair of qkv_post_fwd_kernel written three ways.
    formB        b = fma(x1, c, x0 s): hipcc 7.2 -O3 compiles it to `v_pk_mul_f32 .. op_sel_hi:[0,1] neg_hi:[1,0]` -- the failing build
    formBscalar  the same roundings behind an asm barrier on the products: no packed instruction, clean
    oldrot       the expression of rounds 1-5 (q0 c - q1 s, q1 c + q0 s under -ffp-contract=fast), clean
-> tools/ab/lib/libe2k_<name>.so (E2K_LIB selects one; tools/gpu/r06t.sh runs the probe on them)"""
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
SRC = ROOT / 'e2-tts-pytorch_amd' / 'csrc' / 'attn.hip'
CALLS = '''                rot_pair(q[2 * j], q[2 * j + 1], c, s);
                rot_pair(k[2 * j], k[2 * j + 1], c, s);'''
FORMS = {
    'formB': '''                { const float a_ = fmaf(q[2 * j], c, -(q[2 * j + 1] * s)), b_ = fmaf(q[2 * j + 1], c, q[2 * j] * s); q[2 * j] = a_; q[2 * j + 1] = b_; }
                { const float a_ = fmaf(k[2 * j], c, -(k[2 * j + 1] * s)), b_ = fmaf(k[2 * j + 1], c, k[2 * j] * s); k[2 * j] = a_; k[2 * j + 1] = b_; }''',
    'formBscalar': '''                { float t0 = q[2 * j + 1] * s, t1 = q[2 * j] * s; asm volatile("" : "+v"(t0), "+v"(t1)); const float a_ = fmaf(q[2 * j], c, -t0), b_ = fmaf(q[2 * j + 1], c, t1); q[2 * j] = a_; q[2 * j + 1] = b_; }
                { float t0 = k[2 * j + 1] * s, t1 = k[2 * j] * s; asm volatile("" : "+v"(t0), "+v"(t1)); const float a_ = fmaf(k[2 * j], c, -t0), b_ = fmaf(k[2 * j + 1], c, t1); k[2 * j] = a_; k[2 * j + 1] = b_; }''',
    'oldrot': '''                float q0 = q[2 * j], q1 = q[2 * j + 1], k0 = k[2 * j], k1 = k[2 * j + 1];
                q[2 * j] = q0 * c - q1 * s;  q[2 * j + 1] = q1 * c + q0 * s;
                k[2 * j] = k0 * c - k1 * s;  k[2 * j + 1] = k1 * c + k0 * s;''',
}


def main():
    text = SRC.read_text()
    assert text.count(CALLS) == 1
    specs = []
    with tempfile.TemporaryDirectory() as td:
        for name, body in FORMS.items():
            f = Path(td) / f'attn_{name}.hip'
            f.write_text(text.replace(CALLS, body))
            specs.append(f'{name}={f}:attn.hip')
        subprocess.run([sys.executable, str(ROOT / 'tools' / 'ab' / 'build_variants.py'), *specs], check=True)


if __name__ == '__main__':
    main()
