#!/usr/bin/env python3
"""Disassemble one gfx950 kernel out of an object / shared library and print its instruction mix (developer tool).
  python tools/disasm_kernel.py nnest_amd/csrc/nnest_quad.o 'mh_kernel_quad<2, false>' [--dump out.s]"""
import collections
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from kernel_resources import code_objects  # noqa: E402

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def main():
    path, want = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index('--dump') + 1] if '--dump' in sys.argv else None
    data = open(path, 'rb').read()
    for co in code_objects(data):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([OBJDUMP, '-d', '--demangle', f.name], capture_output=True, text=True).stdout
        for m in re.finditer(r'^[0-9a-f]+ <(.*?)>:\n(.*?)(?=^\n|\Z)', txt, re.S | re.M):
            name, body = m.group(1), m.group(2)
            if want not in name or name.endswith('.kd'):
                continue
            lines = [l for l in body.split('\n') if l.strip() and not l.strip().startswith('<')]
            ops = collections.Counter()
            for l in lines:
                t = l.split('//')[0].split()
                if t:
                    ops[t[0]] += 1
            print(name, len(lines), 'instructions')
            groups = collections.Counter()
            for op, n in ops.items():
                key = ('mfma' if 'mfma' in op else 'scratch' if op.startswith('scratch') else 'accvgpr' if 'accvgpr' in op else
                       'permlane' if 'permlane' in op else 'dpp' if 'dpp' in op else 'ds' if op.startswith('ds_') else
                       'global/flat' if op.startswith(('global', 'flat', 'buffer')) else 's_nop' if op == 's_nop' else
                       's_waitcnt' if op == 's_waitcnt' else 's_barrier' if op == 's_barrier' else
                       'trans' if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', op) else 'valu' if op.startswith('v_') else 'salu')
                groups[key] += n
            print('  ', dict(groups))
            if dump:
                open(dump, 'w').write('\n'.join(lines))
            return


if __name__ == '__main__':
    main()
