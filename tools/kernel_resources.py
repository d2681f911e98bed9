#!/usr/bin/env python3
"""Registers / spills / scratch / LDS of every gfx950 kernel in an object file or shared library (developer tool).
  python tools/kernel_resources.py nnest_amd/libnnest_hip.so [name-substring]
Finds the clang offload bundles inside the file, extracts the gfx950 code objects and reads their metadata notes."""
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'


def code_objects(data):
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from('<Q', data, pos + len(MAGIC))
        off = pos + len(MAGIC) + 8
        for _ in range(n):
            o, sz, tl = struct.unpack_from('<QQQ', data, off)
            triple = data[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if 'gfx950' in triple and sz:
                yield data[pos + o:pos + o + sz]
        pos += len(MAGIC)


def main():
    path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
    data = open(path, 'rb').read()
    for co in code_objects(data):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, '--notes', f.name], capture_output=True, text=True).stdout
        for blk in notes.split('- .agpr_count')[1:]:
            blk = '.agpr_count' + blk
            g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
            name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
            if pat in name:
                print('%-100s vgpr %s agpr %s sgpr %s vspill %s sspill %s scratch %s lds %s' % (
                    name[:100], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('vgpr_spill_count'),
                    g('sgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))


if __name__ == '__main__':
    main()
