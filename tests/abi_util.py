"""Parses include/mvf_hip.h into {function: kind-string} (p pointer, i int, l long, z size_t, f float)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_signatures(path=os.path.join(ROOT, 'include', 'mvf_hip.h')):
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|size_t)\s+(mvf_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        name, args = m.group(1), m.group(2)
        kinds = ''
        for a in [x.strip() for x in args.replace('\n', ' ').split(',') if x.strip()]:
            if '*' in a or 'hipStream_t' in a:
                kinds += 'p'
            elif re.search(r'\buint64_t\b', a):
                kinds += 'u'
            elif re.search(r'\bsize_t\b', a):
                kinds += 'z'
            elif re.search(r'\blong\b', a):
                kinds += 'l'
            elif re.search(r'\bfloat\b', a):
                kinds += 'f'
            elif re.search(r'\bint\b', a):
                kinds += 'i'
            else:
                raise ValueError('cannot classify %r in %s' % (a, name))
        out[name] = kinds
    return out
