"""Disassemble the gfx950 code objects inside a built libevoke_hip*.so and list every packed-f32 VALU instruction whose LOW result reads the
HIGH half of a VGPR operand (v_pk_{add,mul,fma}_f32 ... op_sel:[..1..]).  That form misbehaved on MI355X (csrc/gemm.hip, gate statistics:
a stale high half in lanes 48-63, about once per 10^6 results); the kernels are written so that the compiler does not emit it, and
tests/test_abi.py runs this check on every build.  usage: python tools/check_packed_opsel.py [lib.so ...]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(path):
    """the gfx950 ELF images of every clang offload bundle embedded in the shared library"""
    blob = open(path, 'rb').read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return out
        n, = struct.unpack_from('<Q', blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if 'gfx950' in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)


PK = re.compile(r'\b(v_pk_(?:add|mul|fma)_f32)\s+(.*)')


def suspicious(path):
    """[(kernel, instruction text)] for packed-f32 ops with an op_sel bit set on a VGPR source"""
    bad = []
    for img in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', f.name], capture_output=True, text=True, check=True).stdout
        kern = '?'
        for line in txt.split('\n'):
            m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
            if m:
                kern = m.group(1)
                continue
            m = PK.search(line)
            if not m:
                continue
            text = m.group(2).split('//')[0].strip()
            sel = re.search(r'op_sel:\[([01,]+)\]', text)
            if not sel:
                continue
            ops = [t.strip() for t in text.split(' op_sel')[0].split(',')]
            # operands: dst, src0, src1[, src2]; register pairs print as v[a:b] / s[a:b]
            srcs, i = [], 0
            joined = text.split(' op_sel')[0]
            srcs = re.findall(r'(v\[\d+:\d+\]|s\[\d+:\d+\]|v\d+|s\d+|-?\d+(?:\.\d+)?|0x[0-9a-f]+|vcc|exec)', joined)[1:]
            bits = sel.group(1).split(',')
            if any(b == '1' and k < len(srcs) and srcs[k].startswith('v') for k, b in enumerate(bits)):
                bad.append((kern, m.group(1) + ' ' + text))
    return bad


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = sys.argv[1:] or [os.path.join(here, 'evoke_amd', n) for n in ('libevoke_hip.so', 'libevoke_hip_bf16.so')]
    rc = 0
    for lib in libs:
        n = len(code_objects(lib))
        bad = suspicious(lib)
        print('%s: %d gfx950 code objects, %d packed-f32 instructions with a cross-half low read' % (os.path.basename(lib), n, len(bad)))
        for k, t in bad[:20]:
            print('   ', k[:70], '|', t)
        rc |= 1 if (bad or n == 0) else 0
    sys.exit(rc)
