#!/usr/bin/env python3
"""Static instruction census of the kernels in one object of sevennet_amd/csrc/build (gfx950 code object, llvm-objdump):
    python tools/isa_census.py convf_22d6a77ad5ac [kernel-name regex] [--top 40]
Counts are STATIC (the channel-tile loops of the fused kernels run 8 / 4 / 2 times per x block); they show the instruction mix
the compiler produced -- register copies, hazard no-ops, waits -- not a dynamic profile (that is the SQ counter pass)."""
import argparse
import collections
import os
import re
import subprocess
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('obj')
    ap.add_argument('filt', nargs='?', default='.')
    ap.add_argument('--top', type=int, default=0)
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(root, 'sevennet_amd', 'csrc', 'build' + os.environ.get('SNET_BUILD_SFX', ''), a.obj + '.o')
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([f'{LLVM}/llvm-objcopy', f'--dump-section=.hip_fatbin={d}/fat.bin', obj])
        subprocess.check_call([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={d}/fat.bin',
                               '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={d}/dev.co'])
        text = subprocess.check_output([f'{LLVM}/llvm-objdump', '-d', f'{d}/dev.co']).decode().split('\n')
        names = subprocess.check_output(['c++filt'], input='\n'.join(
            m.group(1) for m in (re.match(r'^[0-9a-f]+ <(.*)>:', x) for x in text) if m).encode()).decode().split('\n')
    starts = [i for i, x in enumerate(text) if re.match(r'^[0-9a-f]+ <', x)]
    cats = [('matrix (v_mfma)', r'v_mfma'), ('packed fp32 (v_pk_*)', r'v_pk_'), ('fp32 fma / mul / add', r'v_(fma|fmac|fmaak|fmamk|mul|add|sub)_f32'),
            ('conversions, bit packing (operand split)', r'v_(cvt|perm_b32|and|lshl|lshr|or|bfe|bfi|pack|ldexp)'),
            ('register copies (v_mov, v_accvgpr)', r'v_(mov|accvgpr)'), ('lane exchange (permlane, dpp, bpermute, readlane)', r'(v_permlane|ds_bpermute|ds_swizzle|v_readlane|v_readfirstlane|v_writelane)'),
            ('other vector', r'v_'), ('LDS reads', r'ds_(read|load)'), ('LDS writes', r'ds_(write|store)'),
            ('global / buffer loads', r'(global|buffer)_load'), ('global / buffer stores', r'(global|buffer)_store'),
            ('s_waitcnt', r's_waitcnt'), ('s_nop (hazard padding)', r's_nop'), ('s_barrier', r's_barrier'), ('other scalar', r's_')]
    for k, i in enumerate(starts):
        name = names[k] if k < len(names) else text[i]
        short = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')
        if not re.search(a.filt, short):
            continue
        j = starts[k + 1] if k + 1 < len(starts) else len(text)
        c = collections.Counter()
        for x in text[i + 1:j]:
            m = re.match(r'^\s+([a-z_0-9]+)\s', x)
            if m:
                c[m.group(1)] += 1
        tot = sum(c.values())
        print(f'{short}: {tot} instructions')
        left = dict(c)
        for label, pat in cats:
            ks = [o for o in left if re.match(pat, o)]
            n = sum(left.pop(o) for o in ks)
            if n:
                print(f'    {label:52s} {n:6d}  {100 * n / tot:5.1f} %')
        if left:
            print(f'    {"other":52s} {sum(left.values()):6d}  {100 * sum(left.values()) / tot:5.1f} %')
        if a.top:
            print('    top: ' + ' '.join(f'{o}:{n}' for o, n in c.most_common(a.top)))


if __name__ == '__main__':
    main()
