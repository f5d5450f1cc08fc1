"""CPU: the spectral-stage kernel keeps its working set in registers -- NO scratch (spill) memory in any instantiation of
k_stft (the s16 and float64 ingest variants spilled 20-108 bytes per lane; only the float32 headline kernel
was clean).  Parsed from the compiler's own resource remarks for the flags the library is built with; hipcc cross-compiles
gfx950 without a GPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resource_usage(src, extra):
    from audfprint_amd import build
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not found')
    cmd = [hipcc] + build.COMMON + list(extra) + ['--cuda-device-only', '-c', os.path.join(build.CSRC, src), '-o', os.devnull,
                                                  '-Rpass-analysis=kernel-resource-usage']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res, cur = {}, None
    for ln in out.stderr.splitlines():
        m = re.search(r'Function Name: (\S+)', ln)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r'remark: .*?\s+(VGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)', ln)
        if m and cur is not None:
            cur[m.group(1).split()[0]] = int(m.group(2))
    return res


def test_k_stft_has_no_scratch_in_any_instantiation():
    from audfprint_amd import build
    flags = next(e[1] for e in build.SOURCES if e[0] == 'k_stft.hip')
    res = _resource_usage('k_stft.hip', flags)
    stft = {k: v for k, v in res.items() if k.startswith('_Z6k_stftI')}
    # ST in {short, float, double} x {dense, compact, dense-from-list}
    assert len(stft) == 9, sorted(res)
    for name, r in sorted(stft.items()):
        assert r['ScratchSize'] == 0, (name, r)
        assert r['VGPRs'] <= 128, (name, r)               # four wavefronts per SIMD (3 beside the scan kernel)
    compact = [k for k in stft if 'Lb1ELb0E' in k]
    assert len(compact) == 3
    for name in compact:
        assert stft[name]['LDS'] <= 40 * 1024, (name, stft[name])     # four workgroups per CU
