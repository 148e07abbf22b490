#!/usr/bin/env python3
"""Parse the POD structs of include/gan_amd.h and emit their ctypes mirror - the binding shown in INTEGRATION.md
is this script's output, and tests/test_cpu_host.py checks header == gan_amd/_lib.py == INTEGRATION.md with it."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CT = {'int32_t': 'C.c_int32', 'uint32_t': 'C.c_uint32', 'float': 'C.c_float', 'size_t': 'C.c_size_t', 'int64_t': 'C.c_int64'}


def parse_structs(header=None):
    """-> {struct name: [(field, ctypes type name)]} in declaration order."""
    text = open(header or os.path.join(ROOT, 'include', 'gan_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    out = {}
    for m in re.finditer(r'typedef struct (\w+) \{(.*?)\} \1;', text, flags=re.S):
        fields = []
        for decl in m.group(2).split(';'):
            decl = ' '.join(decl.split())
            if not decl:
                continue
            mm = re.match(r'(const )?(\w+)\s*(\*?)\s*(.*)$', decl)
            base, ptr, names = mm.group(2), mm.group(3), mm.group(4)
            for nm in [n.strip() for n in names.split(',')]:
                star = ptr or ('*' if nm.startswith('*') else '')
                nm = nm.lstrip('* ')
                ct = 'C.c_void_p' if star else (CT.get(base) or base)       # struct members keep their struct name
                fields.append((nm, ct))
        out[m.group(1)] = fields
    return out


def ctypes_source(names):
    structs = parse_structs()
    lines = []
    for n in names:
        fl = ', '.join(f'("{f}", {t})' for f, t in structs[n])
        lines.append(f'class {n}(C.Structure):            # include/gan_amd.h: {n}\n    _fields_ = [{fl}]\n')
    return '\n'.join(lines)


if __name__ == '__main__':
    print(ctypes_source(sys.argv[1:] or ['GanTensor', 'GanConvDesc']))
