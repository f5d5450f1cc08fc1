"""Build libafp_hip.so (gfx950) in-tree with hipcc.  `python -m audfprint_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', '_obj')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libafp_hip.so')

# (source, extra flags).  k_scan must not contract a*b+c into FMA: the HPF / threshold
# recurrences have to round like the reference's separate numpy operations.
SOURCES = [
    ('k_stft.hip', ['-DSTFT_MINW=%s' % os.environ.get('AFP_STFT_MINW', '3'), '-fno-honor-nans']),
    ('k_scan.hip', ['-ffp-contract=off', '-fno-honor-nans']),
    # the same file again: k_scan_small, the 8 KB-of-LDS scan that leaves room for a third k_stft workgroup per CU
    ('k_scan.hip', ['-ffp-contract=off', '-fno-honor-nans', '-DSCAN_SMALL_LDS=1'], 'k_scan_small.o'),
    ('k_pair.hip', []),
    ('k_table.hip', []),
    ('afp_abi.hip', []),
]
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found; cannot build libafp_hip.so')
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'afp.h'))
    srcs = [os.path.join(CSRC, ent[0]) for ent in SOURCES]
    if not force and os.path.exists(LIB) and not any(_newer(f, LIB) for f in srcs + headers):
        return LIB                                  # library is newer than every source: nothing to do
    objs = []
    relink = force or not os.path.exists(LIB)
    for ent in SOURCES:
        src, extra = ent[0], ent[1]
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, ent[2] if len(ent) > 2 else src.replace('.hip', '.o'))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(hd, o) for hd in headers):
            cmd = [hipcc] + COMMON + extra + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
    if relink:
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
