"""CPU, build container only: the UNCHANGED reference CLI module (audfprint.py) running on top of the
drop-in module -- `setup_analyzer`, `file_precompute`, `ingest`/`make_ht_from_list` plumbing, the
`.afpt` writer and the reference HashTable -- with the GPU calls replaced by canned results (there is
no GPU here; the GPU side of the same calls is covered by tests/test_gpu_dropin.py)."""
import os
import sys
import types

import numpy as np
import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not mounted')


@pytest.fixture()
def ref_cli(monkeypatch):
    import audfprint_amd.audfprint_analyze as M
    monkeypatch.setitem(sys.modules, 'audfprint_analyze', M)          # the drop-in, under the reference's name
    docopt = types.ModuleType('docopt')                               # not installed here; only imported by main()
    docopt.docopt = lambda *a, **k: {}
    monkeypatch.setitem(sys.modules, 'docopt', docopt)
    monkeypatch.syspath_prepend(REF)
    for m in ('audfprint', 'audfprint_match', 'hash_table', 'audio_read', 'stft'):
        sys.modules.pop(m, None)
    import audfprint                                                   # the reference CLI module, unchanged
    yield audfprint, M
    for m in ('audfprint', 'audfprint_match', 'hash_table', 'audio_read', 'stft'):
        sys.modules.pop(m, None)


ARGS = {'--density': '70', '--pks-per-frame': '5', '--fanout': '8', '--freq-sd': '30.0', '--shifts': '0',
        '--samplerate': '11025', '--continue-on-error': True, 'match': True}


def test_setup_analyzer_builds_the_dropin(ref_cli):
    audfprint, M = ref_cli
    a = audfprint.setup_analyzer(ARGS)
    assert isinstance(a, M.Analyzer)
    assert (a.density, a.maxpksperframe, a.maxpairsperpeak, a.f_sd, a.shifts) == (70.0, 5, 8, 30.0, 4)
    assert (a.target_sr, a.n_fft, a.n_hop, a.fail_on_error) == (11025, 512, 256, False)


def test_precompute_and_new_plumbing(ref_cli, tmp_path, monkeypatch):
    audfprint, M = ref_cli
    from conftest import load_golden
    g = load_golden('noise_s0_10s')
    a = audfprint.setup_analyzer(dict(ARGS, **{'--density': '20', '--fanout': '3', 'match': False}))

    def fake_wavfile2hashes(self, filename):            # stands in for the GPU call
        self.soundfiledur = 10.0
        self.soundfiletotaldur += 10.0
        self.soundfilecount += 1
        return g['hashes']
    monkeypatch.setattr(M.Analyzer, 'wavfile2hashes', fake_wavfile2hashes)
    msgs = audfprint.file_precompute(a, './some/dir/clip.wav', str(tmp_path), type='hashes')
    out = tmp_path / 'some' / 'dir' / 'clip.afpt'
    assert out.exists() and 'wrote' in msgs[0] and '669 hashes' in msgs[0]
    assert open(str(out), 'rb').read() == b'audfprinthashV00' + g['hashes'].astype('<i4').tobytes()
    # skip-existing, and the reference's own table built through Analyzer.ingest (audfprint.py:130-144)
    assert 'skipping' in audfprint.file_precompute(a, './some/dir/clip.wav', str(tmp_path), type='hashes', skip_existing=True)[0]
    ht = audfprint.make_ht_from_list(a, ['x.wav', 'y.wav'], 20, 100, 16384)
    assert ht.names == ['x.wav', 'y.wav'] and int(ht.counts.sum()) == 2 * len(g['hashes'])
    assert a.soundfilecount == 3
    # the .afpt written above short-circuits wavfile2hashes of a fresh Analyzer (no GPU involved)
    monkeypatch.undo()                                   # (also removes the module aliases again)
    b = M.Analyzer()
    back = b.wavfile2hashes(str(out))
    assert np.array_equal(np.array(back), g['hashes']) and b.soundfilecount == 1
    assert sys.modules.get('audfprint_analyze') is not M
