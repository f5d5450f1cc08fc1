"""the host machinery of the pipelined ingest (upload stream, staged contexts, device table store with
overflow replay, sparse / dense table download through the pinned ring and the host-thread pool, the one-file export path)
under sustained load -- tools/soak.py, 200 iterations here (the committed profiles/r05_soak*.log hold the 2 000-iteration
runs), every iteration's results compared with the first's, in a child process so that a GPU memory fault is a test
failure with the native stack on stderr, not the end of the test session."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['torch', 'notorch'])
def test_soak_200_iterations(mode):
    env = dict(os.environ, AFP_BACKTRACE='1', PYTHONFAULTHANDLER='1')
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'soak.py'), '--iters', '200', '--clips', '2500', '--c3-clips', '256', '--tag', 'suite-' + mode]
    if mode == 'notorch':
        cmd.append('--no-torch')            # the system HIP runtime, pinned memory from afp_pinned_alloc
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    tail = (out.stdout[-3000:] + '\n' + out.stderr[-3000:])
    assert out.returncode == 0, tail
    assert 'DONE' in out.stdout and '"mismatches": 0' in out.stdout and '"iters_done": 200' in out.stdout, tail
