"""Builds libmvf_hip.so (gfx950) in-tree with hipcc: every .hip in this directory is compiled to an object
(in parallel) and linked into one shared library next to the sources.  No JIT cache: the .so travels with
the tree.  Usage: python -m video_rep_learning_amd.csrc.build [--force]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libmvf_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# atomic optimizer off: it rewrites a one-lane ticket atomic (gemm_tc256) into a wave-aggregated one whose result is
# read back at once (s_waitcnt vmcnt(0) right behind the atomic)
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result',
         '-mllvm', '-amdgpu-atomic-optimizer-strategy=None',
         '-I' + HERE, '-I' + os.path.join(HERE, '..', '..', 'include')] + os.environ.get('MVF_EXTRA_FLAGS', '').split()


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.hip'))


def headers():
    inc = os.path.join(HERE, '..', '..', 'include')
    return [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')] + \
           [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h')]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(HERE, 'build', os.path.basename(src)[:-4] + '.o')
    if force or _stale(obj, [src] + headers()):
        r = subprocess.run([HIPCC] + FLAGS + ['-c', src, '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, r.stderr[-4000:]))
    return obj


def build(force=False, verbose=False):
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), sources()))
    if force or _stale(LIB, objs):
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
        if verbose:
            print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
