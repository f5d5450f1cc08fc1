"""GPU: staged mode (afp_set_stage_streams).  Handles that share a spectral-stage stream and a scan-stage
stream pipeline consecutive batches against each other; the rows they return must be the very rows the
single-stream path and the oracle give, whatever the interleaving."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(seed, n, secs):
    from oracle import afp_oracle as O
    return [O.synth_noise(seed + i, secs) if i % 3 else O.synth_tonal(seed + i, secs) for i in range(n)]


@pytest.mark.parametrize('nstages', [2, 3])
def test_staged_pipeline_matches_single_stream_and_oracle(nstages):
    import torch
    from oracle import afp_oracle as O
    from audfprint_amd.batch import Extractor
    dev = torch.device('cuda', 0)
    exs = [Extractor(0) for _ in range(3)]
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
    sc = torch.cuda.Stream(device=dev, priority=-1) if nstages == 3 else None
    for e in exs:
        e.set_params(density=20.0)
        e.set_stage_streams(sa.cuda_stream, sb.cuda_stream, sc.cuda_stream if sc else None)
    # distinct ragged batches, resident in HBM
    batches = []
    for b in range(6):
        clips = _mk(9000 + 37 * b, 5 + b, 2.0 + 0.7 * b)
        clips.insert(1, np.zeros(0, np.float32))                     # an empty clip inside the batch
        pcm, off = Extractor.pack(clips, np.float32)
        batches.append((clips, torch.from_numpy(pcm).to(dev), off))
    torch.cuda.synchronize()
    # three batches in flight at any time
    got = [None] * len(batches)
    inflight = []
    for b, (clips, d, off) in enumerate(batches):
        e = exs[b % len(exs)]
        if len(inflight) == len(exs):
            pb, pe = inflight.pop(0)
            got[pb] = pe.fetch(len(batches[pb][0]), True, True)
        e.extract_device(d.data_ptr(), off, want_hashes=True, want_peaks=True)
        inflight.append((b, e))
    for pb, pe in inflight:
        got[pb] = pe.fetch(len(batches[pb][0]), True, True)
    # reference run: one handle, one stream
    ref = Extractor.get(0)
    ref.set_params(density=20.0)
    for b, (clips, d, off) in enumerate(batches):
        ref.extract_device(d.data_ptr(), off, want_hashes=True, want_peaks=True)
        r = ref.fetch(len(clips), True, True)
        g = got[b]
        assert np.array_equal(g.hashes, r.hashes) and np.array_equal(g.hash_offsets, r.hash_offsets), b
        assert np.array_equal(g.peaks, r.peaks) and np.array_equal(g.peak_offsets, r.peak_offsets), b
    # and against the oracle for one batch
    clips = batches[2][0]
    for i, c in enumerate(clips):
        pls, hs = O.extract(c, O.Params())
        assert np.array_equal(got[2].clip_hashes(i), hs), i
    # switching back to single-stream mode keeps working
    for e in exs:
        e.set_stage_streams(None, None)
    clips, d, off = batches[1]
    exs[0].extract_device(d.data_ptr(), off, want_hashes=True, want_peaks=True)
    r = exs[0].fetch(len(clips), True, True)
    assert np.array_equal(r.hashes, got[1].hashes)


def test_stage_streams_argument_errors():
    import torch
    from audfprint_amd import _lib
    from audfprint_amd.batch import Extractor
    e = Extractor(0)
    s = torch.cuda.Stream(device=torch.device('cuda', 0))
    with pytest.raises(_lib.AfpError):
        e.set_stage_streams(s.cuda_stream, None)            # both or neither
    with pytest.raises(_lib.AfpError):
        e.set_stage_streams(s.cuda_stream, s.cuda_stream)   # must be two different streams
    with pytest.raises(_lib.AfpError):
        e.set_stage_streams(None, None, s.cuda_stream)      # a pair stage needs the other two


@pytest.mark.parametrize('mode', ['small', 'big'])
def test_scan_lds_variants_equal_oracle(mode, monkeypatch):
    """k_scan exists twice: the 2-frame-ring kernel and the 8 KB-of-LDS kernel (1-frame ring slots, last
    column and backward record ring parked in idle ring space) that large batches use.  Both must give the
    oracle's rows, in particular for units of 1, 2, 3, 4, 5 frames where the slot parity logic is exercised."""
    from oracle import afp_oracle as O
    from audfprint_amd.batch import Extractor
    monkeypatch.setenv('AFP_SCAN_LDS', mode)
    e = Extractor(0)
    monkeypatch.delenv('AFP_SCAN_LDS')
    clips = [O.synth_noise(31, 30.0)[:n] for n in (100, 256, 700, 900, 1100, 1300, 2000, 4000, 11025, 60000)]
    clips += [O.synth_tonal(33, 6.0), O.synth_noise(34, 12.5), np.zeros(3000, np.float32)]
    for kw in (dict(), dict(density=70.0, maxpairsperpeak=10, shifts=4), dict(maxpksperframe=1), dict(maxpksperframe=8)):
        e.set_params(**kw)
        r = e.extract(clips=clips, want_hashes=True, want_peaks=True)
        prm = O.Params(**kw)
        for i, c in enumerate(clips):
            pls, hs = O.extract(c, prm)
            assert np.array_equal(r.clip_hashes(i), hs), (mode, kw, i)
            for s in range(prm.shifts):
                assert np.array_equal(r.unit_peaks(i, s), pls[s]), (mode, kw, i, s)
    e.close()
