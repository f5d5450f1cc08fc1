"""Build libafp_hip.so (gfx950) in-tree with hipcc.  `python -m audfprint_amd.build`.

The library embeds `afp_build_id()` = sha256 of every kernel / ABI source and of the compile flags; a build is
redone whenever that id differs from the one the existing library was linked with (a sidecar file next to the
.so), and `_lib.load()` refuses a library whose embedded id is not the tree's -- a stale binary can neither be
benchmarked nor tested."""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', '_obj' + os.environ.get('AFP_OBJ_SUFFIX', ''))
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.environ.get('AFP_LIB_PATH') or os.path.join(LIBDIR, 'libafp_hip.so')      # (override: A/B builds of variants)
IDFILE = os.path.splitext(LIB)[0] + '.build_id'

# (source, extra flags).  k_scan must not contract a*b+c into FMA: the HPF / threshold
# recurrences have to round like the reference's separate numpy operations.
# per-file extra flags for A/B builds of one kernel file (part of the build id like everything else)
STFT_X = os.environ.get('AFP_STFT_FLAGS', '').split()
SCAN_X = os.environ.get('AFP_SCAN_FLAGS', '').split()
SOURCES = [
    ('k_stft.hip', ['-DSTFT_MINW=%s' % os.environ.get('AFP_STFT_MINW', '4'), '-fno-honor-nans'] + STFT_X),
    ('k_scan.hip', ['-ffp-contract=off', '-fno-honor-nans', '-Wno-inline-asm'] + SCAN_X),
    # the same file again: k_scan_small, the 8 KB-of-LDS scan that leaves room for a third k_stft workgroup per CU
    ('k_scan.hip', ['-ffp-contract=off', '-fno-honor-nans', '-Wno-inline-asm', '-DSCAN_SMALL_LDS=1'] + SCAN_X, 'k_scan_small.o'),
    ('k_pair.hip', []),
    ('k_table.hip', []),
    # host side of the C ABI: handle + pipeline / table + matcher / host pool, buffers, probes (one header: afp_internal.h)
    ('afp_abi.hip', []),
    ('afp_table.hip', []),
    ('afp_host.hip', []),
]
# -fvisibility=hidden: the dynamic symbol table holds exactly the AFP_API functions of include/afp.h -- the launchers the
# translation units call each other through (afp_launch_*) stay inside the library (tests/test_abi_cpu.py)
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-Wall', '-Wno-unused-function']
EXTRA = os.environ.get('AFP_EXTRA_HIPCC_FLAGS', '').split()      # (A/B builds of kernel variants)


MAPFILE = os.path.join(CSRC, 'libafp.map')


def _headers():
    """everything besides the sources that decides what the library is: the headers and the linker version script"""
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h') or f.endswith('.map'))
    hs.append(os.path.join(HERE, '..', 'include', 'afp.h'))
    return hs


def source_id():
    """First 16 hex digits of sha256 over the sources, headers and flags; None if the sources are not there."""
    h = hashlib.sha256()
    try:
        for ent in SOURCES:
            h.update(repr((ent[0], ent[1], ent[2:] and ent[2])).encode())
        h.update(repr(COMMON + EXTRA).encode())
        for f in sorted(set(os.path.join(CSRC, ent[0]) for ent in SOURCES)) + _headers():
            with open(f, 'rb') as fh:
                h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def _obj_key(cmd, src, headers):
    """sha256 over the exact compile command, the source and every header: an object is reused only if all three match
    (mtimes alone let an A/B build with other flags leave a newer, differently compiled object behind)."""
    h = hashlib.sha256(repr(cmd).encode())
    for f in [src] + list(headers):
        with open(f, 'rb') as fh:
            h.update(b'\0' + os.path.basename(f).encode() + b'\0' + fh.read())
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _built_id():
    return _read(IDFILE)


def build(force=False, verbose=True):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    sid = source_id()
    if not force and os.path.exists(LIB) and sid is not None and _built_id() == sid:
        return LIB                                  # the library was linked from exactly these sources
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found; cannot build libafp_hip.so')
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = _headers()
    objs = []
    for ent in SOURCES:
        src, extra = ent[0], list(ent[1]) + EXTRA
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, ent[2] if len(ent) > 2 else src.replace('.hip', '.o'))
        objs.append(o)
        is_abi = src == 'afp_abi.hip'
        if is_abi:
            extra = extra + ['-DAFP_BUILD_ID="%s"' % sid]
        cmd = [hipcc] + COMMON + extra + ['-c', s, '-o', o]
        key = _obj_key(cmd, s, headers)
        if force or not os.path.exists(o) or _read(o + '.key') != key:
            if verbose:
                print(' '.join(cmd), flush=True)
            if os.path.exists(o + '.key'):
                os.remove(o + '.key')
            subprocess.check_call(cmd)
            with open(o + '.key', 'w') as f:
                f.write(key + '\n')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--version-script=' + MAPFILE, '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(IDFILE, 'w') as f:
        f.write(sid + '\n')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB, source_id())
