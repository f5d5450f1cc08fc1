"""CPU: ADVICE r2 -- k_stft issues its exchange reads as hand-written `ds_read_b64` (inline asm, outside the compiler's
lgkmcnt tracking): the destination registers hold garbage until the `s_waitcnt lgkmcnt(0)` of lds_wait8.  Nothing in the
source orders the two but data flow, so this test reads the generated ISA of every k_stft variant and fails if any
instruction between a read block and its wait touches one of the registers in flight (a spill or a copy the register
allocator might one day insert)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_no_use_of_lds_read_destinations_before_the_wait(tmp_path):
    from audfprint_amd import build as B
    src = os.path.join(B.CSRC, 'k_stft.hip')
    flags = next(e[1] for e in B.SOURCES if e[0] == 'k_stft.hip')
    out = str(tmp_path / 'k_stft.s')
    subprocess.check_call([HIPCC] + B.COMMON + list(flags) + ['-S', '--cuda-device-only', src, '-o', out])
    lines = open(out).read().split('\n')
    nblocks = 0
    inflight = set()
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        op = t.split()[0]
        if op == 'ds_read_b64' and 'offset' in t and re.match(r'ds_read_b64\s+v\[\d+:\d+\],', t):
            # (the hand-issued reads; the compiler's own ds_read_b64 are tracked by its waitcnt insertion and also end
            #  at the next lgkmcnt(0) wait, so treating them alike is conservative)
            inflight |= _regs(t.split(',')[0])
            nblocks += 1
            continue
        if op == 's_waitcnt' and 'lgkmcnt(0)' in t:
            inflight.clear()
            continue
        if op.startswith('s_') or not inflight:
            continue
        # any other instruction: none of its operands may be a register still in flight
        ops = t[len(op):]
        used = _regs(ops)
        assert not (used & inflight), 'register in flight touched before its wait: %r (in flight: %s)' % (t, sorted(used & inflight))
    assert nblocks >= 6 * 32, 'the hand-issued reads were not found (%d)' % nblocks
