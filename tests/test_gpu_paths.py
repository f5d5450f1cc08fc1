"""GPU: the three kernel paths of the extraction -- dense (k_stft -> k_scan), COMPACT (k_stft<ST,true> -> k_scan_c: the
log-spectrogram never reaches HBM) and SEGMENT-parallel scan (k_hpf -> k_scan_seg) -- must give the reference's integers on
every input; forced through afp_set_pipeline so each is exercised whatever the batch size."""
import numpy as np
import pytest

from conftest import SPARSE_FRAME, golden_names, load_golden

pytestmark = pytest.mark.gpu

PKEYS = ('density', 'maxpksperframe', 'maxpairsperpeak', 'f_sd', 'shifts', 'targetdf', 'mindt', 'targetdt')
PATHS = {'dense': dict(compact=0, seg=0), 'compact': dict(compact=1, seg=0), 'segments': dict(compact=0, seg=1),
         'segments_short_warmup': dict(compact=0, seg=1, seg_len=16, seg_warm=4)}


@pytest.fixture(scope='module')
def ex():
    from audfprint_amd.batch import Extractor
    e = Extractor.get(0)
    yield e
    e.set_pipeline()            # defaults back for the other test modules


def _params_of(g):
    p = dict(g['params'])
    return p


@pytest.mark.parametrize('path', sorted(PATHS))
def test_every_golden_on_every_path(ex, path):
    """All golden fixtures (generated from the live reference) as ONE mixed batch per parameter set, on a forced path."""
    from audfprint_amd import _lib
    ex.set_pipeline(**PATHS[path])
    groups = {}
    for name in golden_names():
        g = load_golden(name)
        groups.setdefault(tuple(sorted((k, g['params'][k]) for k in PKEYS)), []).append((name, g))
    assert groups
    for key, items in groups.items():
        ex.set_params(**dict(key))
        r = ex.extract(clips=[g['d'] for _, g in items], want_hashes=True, want_peaks=True)
        for i, (name, g) in enumerate(items):
            tie = bool(r.unit_flags[i * len(g['peaks'])] & _lib.UNIT_TIE)
            # the flag belongs to the sparse-frame fixtures and to nothing else: a spurious flag on this path must not skip a comparison
            assert tie == (name in SPARSE_FRAME), (path, name, 'AFP_UNIT_TIE %s' % tie)
            if tie:
                continue        # (tests/test_gpu_parity.py and tests/test_gpu_corners.py check that class's own contract)
            for sft, pk in enumerate(g['peaks']):
                assert np.array_equal(r.unit_peaks(i, sft), pk), (path, name, sft)
            assert np.array_equal(r.clip_hashes(i), g['hashes']), (path, name)


def test_long_clips_segmented_vs_oracle(ex):
    """300 s of a tonal + gated signal and a 3600 s clip: segments (default parameters) against the CPU oracle; no segment
    re-run is expected, the final boundary check must pass."""
    from oracle import afp_oracle as O
    ex.set_pipeline(compact=0, seg=1)
    ex.set_params()
    for d in (O.synth_tonal(31, 300.0), np.tile(O.synth_noise(32, 600.0), 6)):
        r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        assert st['used'] and st['segments'] > 20 and not st['failed'], st
        pls, hs = O.extract(d, O.Params())
        assert np.array_equal(r.unit_peaks(0), pls[0]) and np.array_equal(r.clip_hashes(0), hs)


def test_segment_repair_and_fallback_are_exact(ex):
    """A warm-up far too short to converge: the chain launch re-runs the segments (whole runs of them, sequentially, from the
    last true state); with the final check forced to fail the sequential kernel produces the result -- bit-exact either way."""
    from oracle import afp_oracle as O
    ex.set_params()
    d = O.synth_noise(41, 120.0)
    pls, hs = O.extract(d, O.Params())
    seen_rerun = False
    for seg_len, warm, force in ((64, 8, False), (256, 32, False), (128, 2, False), (128, 2, True), (0, 0, True)):
        ex.set_pipeline(compact=0, seg=1, seg_len=seg_len, seg_warm=warm, seg_force_fail=force)
        r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        assert st['used'] and st['failed'] == force, st
        seen_rerun = seen_rerun or st['rerun_fwd'] + st['rerun_bwd'] > 0
        assert np.array_equal(r.unit_peaks(0), pls[0]) and np.array_equal(r.clip_hashes(0), hs), (seg_len, warm, st)
    assert seen_rerun


def test_compact_equals_dense_on_a_ragged_batch(ex):
    """1100 clips of 0.02 .. 12 s (noise, tonal, silence, clipped): compact and dense paths row for row, units that need
    the floor included (they take the dense kernels inside the compact pipeline)."""
    from oracle import afp_oracle as O
    from audfprint_amd import _lib
    rng = np.random.RandomState(7)
    clips = []
    for i in range(1100):
        secs = float(rng.uniform(0.02, 12.0))
        kind = i % 5
        if kind == 0:
            d = O.synth_tonal(2000 + i, max(secs, 0.5))[:int(secs * 11025) + 1]
        elif kind == 1:
            d = np.zeros(int(secs * 11025), np.float32)
        elif kind == 2:
            d = np.clip(O.synth_noise(2000 + i, secs) * 20.0, -1.0, 1.0).astype(np.float32)
        else:
            d = O.synth_noise(2000 + i, secs)
        clips.append(d)
    ex.set_params()
    ex.set_pipeline(compact=0, seg=0)
    a = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    ex.set_pipeline(compact=1, seg=0)
    b = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    assert np.array_equal(a.hash_offsets, b.hash_offsets) and np.array_equal(a.hashes, b.hashes)
    assert np.array_equal(a.peak_offsets, b.peak_offsets) and np.array_equal(a.peaks, b.peaks)
    assert np.array_equal(a.unit_flags, b.unit_flags)
    assert int(np.count_nonzero(a.unit_flags & _lib.UNIT_CORR)) > 0          # the floored path was part of it
    for i in (3, 4, 8, 9, 503, 1099):
        pls, hs = O.extract(clips[i], O.Params())
        assert np.array_equal(b.clip_hashes(i), hs) and np.array_equal(b.unit_peaks(i), pls[0]), i


def _gappy(seed, secs=120.0):
    """Noise with a loud burst, a stretch 30 dB down and 6 s of digital silence: the thresholds of :226-230 remember the
    loud part for hundreds of frames, so segments that start in the quiet stretches cannot converge from their warm-up."""
    from oracle import afp_oracle as O
    d = O.synth_noise(seed, secs).copy()
    sr = 11025
    d[10 * sr:12 * sr] *= 8.0
    d[12 * sr:40 * sr] *= 0.03
    d[60 * sr:66 * sr] = 0.0
    d[66 * sr:67 * sr] *= 6.0
    d[67 * sr:90 * sr] *= 0.1
    return np.clip(d, -1.0, 1.0).astype(np.float32)


def test_segments_over_quiet_stretches(ex):
    """Runs of segments whose warm-up cannot converge: the chain launch re-runs each run sequentially from the last true
    state -- the oracle's result, and no unit left to the sequential kernel."""
    from oracle import afp_oracle as O
    ex.set_params()
    for seed, dens in ((51, 20.0), (52, 70.0)):
        ex.set_params(density=dens)
        d = _gappy(seed)
        pls, hs = O.extract(d, O.Params(density=dens))
        ex.set_pipeline(compact=0, seg=1)
        r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        assert st['used'] and st['segments'] > 20, st
        assert np.array_equal(r.unit_peaks(0), pls[0]) and np.array_equal(r.clip_hashes(0), hs), st
        assert st['rerun_fwd'] + st['rerun_bwd'] > 0, st                  # the signal does break the warm-up premise
        assert not st['failed'], st
    ex.set_params()


def test_segments_in_a_batch_of_long_clips(ex):
    """Several long clips in one batch (some with quiet stretches), segmented: all equal the dense path's; then with the final
    check forced to fail: every unit re-done by the sequential kernel, same result."""
    from oracle import afp_oracle as O
    ex.set_params()
    clips = [O.synth_noise(61, 60.0), _gappy(62, 100.0), O.synth_tonal(63, 45.0), _gappy(64, 95.0), O.synth_noise(65, 30.0)]
    ex.set_pipeline(compact=0, seg=0)
    r0 = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
    h0, p0, o0 = r0.hashes.copy(), r0.peaks.copy(), r0.hash_offsets.copy()
    for force in (False, True):
        ex.set_pipeline(compact=0, seg=1, seg_force_fail=force)
        r1 = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        assert st['used'] and st['failed_units'] == (len(clips) if force else 0), st
        assert np.array_equal(h0, r1.hashes) and np.array_equal(p0, r1.peaks) and np.array_equal(o0, r1.hash_offsets), st


def test_compact_handoff_fault_is_redone_on_the_dense_path(ex):
    """VERDICT r3 #5 / ADVICE r3: the compact stage's hand-off has a recovery.  With the test hook one chunk withholds the
    filter state its successor waits for (and the wait is bounded to ~1 ms): the successor reports a fault, and fetching
    the results re-runs the whole batch on the dense path -- same integers as the oracle, the event counted, and the next
    (healthy) compact batch on the same handle is unaffected.  Host PCM, device-resident PCM and int16 alike."""
    import torch
    from oracle import afp_oracle as O
    from audfprint_amd.batch import Extractor
    ex.set_params()
    clips = [O.synth_noise(900 + i, 5.0 + 0.37 * i) for i in range(12)] + [O.synth_tonal(950, 4.0)]
    want = [O.extract(d, O.Params()) for d in clips]
    before = None

    def check(r, tag):
        for i, (pls, hs) in enumerate(want):
            assert np.array_equal(r.unit_peaks(i, 0), pls[0]) and np.array_equal(r.clip_hashes(i), hs), (tag, i)

    try:
        ex.set_pipeline(compact=1, seg=0)
        r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        st = ex.path_stats()
        assert st['compact'] and not st['redone_dense'], st
        before = st['redone_total']
        check(r, 'healthy compact')
        # fault on a host batch
        ex.set_pipeline(compact=1, seg=0, compact_force_timeout=True)
        r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        st = ex.path_stats()
        assert st['redone_dense'] and not st['compact'] and st['redone_total'] == before + 1, st
        check(r, 'host batch after a forced fault')
        # fault on device-resident int16 PCM (the redo reads the caller's buffer again)
        pcm, off = Extractor.pack([np.round(d * 32768).astype(np.int16) for d in clips], np.int16)
        t = torch.from_numpy(pcm).to('cuda:0')
        torch.cuda.synchronize()
        ex.extract_device(t.data_ptr(), off, want_hashes=True, want_peaks=True, s16=True)
        r = ex.fetch(len(clips), True, True)
        st = ex.path_stats()
        assert st['redone_dense'] and st['redone_total'] == before + 2, st
        check(r, 'device s16 batch after a forced fault')
        # hook off: the same handle runs compact batches again, no redo, flags of the aborted launches do not leak
        ex.set_pipeline(compact=1, seg=0)
        for _ in range(2):
            r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
            st = ex.path_stats()
            assert st['compact'] and not st['redone_dense'] and st['redone_total'] == before + 2, st
            check(r, 'compact again')
    finally:
        ex.set_pipeline()


def test_set_pipeline_none_restores_the_creation_time_selection(ex):
    """set_pipeline() must not wipe a selection made through the environment.  A handle created under
    AFP_COMPACT=1 / AFP_SEG_LEN=16 keeps them across a forced excursion; arguments left None mean 'as created'."""
    import os
    from oracle import afp_oracle as O
    from audfprint_amd.batch import Extractor
    old = {k: os.environ.get(k) for k in ('AFP_COMPACT', 'AFP_SEG')}
    os.environ['AFP_COMPACT'] = '1'
    os.environ['AFP_SEG'] = '0'
    try:
        e2 = Extractor(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        e2.set_params()
        clips = [O.synth_noise(990, 3.0), O.synth_noise(991, 2.0)]
        e2.extract(clips=clips)
        assert e2.path_stats()['compact']                       # two units: only the environment makes this batch compact
        e2.set_pipeline(compact=0, seg=1)
        e2.extract(clips=clips)
        assert not e2.path_stats()['compact']
        e2.set_pipeline()                                       # back to what the environment chose
        e2.extract(clips=clips)
        assert e2.path_stats()['compact'] and not e2.path_stats()['segments']
    finally:
        e2.close()


def _gated_tonal(seed, secs):
    """FM sinusoids under a 2 Hz square gate + a little noise, int16-quantised (tools/seg_cut_sweep.py): loud passages the
    decaying thresholds remember for hundreds of frames -- the input a short warm-up does not converge on."""
    rng = np.random.RandomState(seed)
    n = int(11025 * secs)
    t = np.arange(n) / 11025.0
    x = np.zeros(n)
    for _ in range(12):
        f0, fd, fm = rng.uniform(200, 4500), rng.uniform(0, 80), rng.uniform(0.2, 3)
        x += rng.uniform(0.02, 0.1) * np.sin(2 * np.pi * f0 * t + fd / fm * np.sin(2 * np.pi * fm * t))
    x *= (np.sin(2 * np.pi * 2.0 * t) > 0)
    x += 0.001 * rng.randn(n) * (np.sin(2 * np.pi * 0.25 * t) > -0.5)
    return (np.round(np.clip(x, -1, 1) * 32767).astype(np.int16).astype(np.float32) / np.float32(32768))


def test_short_files_take_the_short_cut_and_back_off_when_it_does_not_converge(ex):
    """VERDICT r4 #5b: a file of up to 1000 frames is cut into segments of 32 + 96 frames instead of 64 + 128 (15 % less per
    call on noise, profiles/r05_seg_cut_sweep.jsonl); a batch that re-ran more than 5 % of its segments sends the handle's
    next batches back to the standard cut.  Bit-exact either way -- the choice is about time only."""
    from oracle import afp_oracle as O
    ex.set_pipeline()
    ex.set_params()
    prm = O.Params()
    d = O.synth_noise(91, 10.0)
    want = O.extract(d, prm)
    r = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
    st = ex.seg_stats()
    assert st['used'] and (st['seg_len'], st['seg_warm']) in ((32, 96), (64, 128)), st
    assert np.array_equal(r.unit_peaks(0, 0), want[0][0]) and np.array_equal(r.clip_hashes(0), want[1])
    b0 = st['short_cut_backoffs']
    # a clip the short warm-up cannot converge on: whatever cut this call takes, the result is the oracle's; if it took the short
    # cut and re-ran > 5 % of its segments the handle backs off, and the NEXT short file takes the standard cut
    g = _gated_tonal(5, 10.0)
    wg = O.extract(g, prm)
    for _ in range(40):                                   # (a back-off left by an earlier test runs out within 32 batches)
        rg = ex.extract(clips=[g], want_hashes=True, want_peaks=True)
        sg = ex.seg_stats()
        assert np.array_equal(rg.unit_peaks(0, 0), wg[0][0]) and np.array_equal(rg.clip_hashes(0), wg[1])
        if (sg['seg_len'], sg['seg_warm']) == (32, 96):
            break
    assert (sg['seg_len'], sg['seg_warm']) == (32, 96), sg
    if (sg['rerun_fwd'] + sg['rerun_bwd']) * 20 > sg['segments']:
        assert sg['short_cut_backoffs'] == b0 + 1 or sg['short_cut_backoffs'] > b0, (sg, b0)
        r2 = ex.extract(clips=[d], want_hashes=True, want_peaks=True)
        s2 = ex.seg_stats()
        assert (s2['seg_len'], s2['seg_warm']) == (64, 128), s2
        assert np.array_equal(r2.unit_peaks(0, 0), want[0][0]) and np.array_equal(r2.clip_hashes(0), want[1])
    # a long file keeps the standard cut
    dl = O.synth_noise(92, 60.0)
    rl = ex.extract(clips=[dl], want_hashes=True, want_peaks=False)
    sl = ex.seg_stats()
    assert sl['used'] and (sl['seg_len'], sl['seg_warm']) == (64, 128), sl
    assert np.array_equal(rl.clip_hashes(0), O.extract(dl, prm)[1])


def test_chunked_onset_filter_of_long_units(ex):
    """Round 6: units of 4096 frames and more filter their chunks in parallel (k_hpf chunk mode: granule end states, entry
    states folded from them, a 1024-frame warm-up, chunk boundaries compared bit for bit) instead of carrying the filter state
    through the whole unit.  Exact by that comparison: the 300 s KAT of SURVEY §8c, a tonal + gated clip, a clip with a long
    digital silence, a batch of three long units of different lengths -- all equal to the oracle with no unit failing the
    check; with the comparison forced to fail the sequential kernel delivers the same rows."""
    from conftest import load_golden
    from oracle import afp_oracle as O
    ex.set_pipeline(compact=0, seg=1)
    ex.set_params()
    g = load_golden('noise_s0_300s')
    n0 = ex.path_stats()['hpf_chunked_total'] if ex.last_nclips else 0
    r = ex.extract(clips=[g['d']], want_hashes=True, want_peaks=True)
    st, ps = ex.seg_stats(), ex.path_stats()
    assert st['used'] and not st['failed'] and ps['segments'] and ps['hpf_chunked_total'] == n0 + 1, (st, ps)
    assert np.array_equal(r.clip_hashes(0), g['hashes']) and np.array_equal(r.unit_peaks(0), g['peaks'][0])
    quiet = np.concatenate([O.synth_noise(61, 60.0), np.zeros(40 * 11025, np.float32), O.synth_noise(62, 60.0)])
    clips = [O.synth_tonal(63, 130.0), quiet, O.synth_noise(64, 96.0 + 1.0 / 3)]
    want = [O.extract(d, O.Params()) for d in clips]
    for batch in ([clips[0]], [clips[1]], clips):
        r = ex.extract(clips=batch, want_hashes=True, want_peaks=True)
        st, ps2 = ex.seg_stats(), ex.path_stats()
        assert st['used'] and not st['failed'] and ps2['hpf_chunked_total'] > ps['hpf_chunked_total'], (st, ps2)
        ps = ps2
        for i, d in enumerate(batch):
            pls, hs = want[[id(c) for c in clips].index(id(d))]
            assert np.array_equal(r.unit_peaks(i), pls[0]) and np.array_equal(r.clip_hashes(i), hs), (len(batch), i)
    # long and short units, an empty one, raw s16 and float64 samples in one batch: every unit is covered by the chunk lists
    mixed = [clips[2], O.synth_noise(66, 5.0), np.zeros(0, np.float32), O.synth_noise(67, 100.0), np.zeros(110 * 11025, np.float32)]
    r = ex.extract(clips=mixed, want_hashes=True, want_peaks=True)
    assert ex.seg_stats()['used'] and not ex.seg_stats()['failed']
    for i, d in enumerate(mixed):
        pls, hs = O.extract(d, O.Params())
        assert np.array_equal(r.unit_peaks(i), pls[0]) and np.array_equal(r.clip_hashes(i), hs), i
    for conv in (lambda d: np.round(d * 32768).astype(np.int16), lambda d: d.astype(np.float64)):
        r = ex.extract(clips=[conv(clips[2])], want_hashes=True, want_peaks=False)
        assert ex.seg_stats()['used'] and not ex.seg_stats()['failed'] and np.array_equal(r.clip_hashes(0), want[2][1])
    ps = ex.path_stats()
    # a unit just under the threshold keeps the sequential filter
    short = O.synth_noise(65, 90.0)                          # 3876 frames
    r = ex.extract(clips=[short], want_hashes=True, want_peaks=False)
    assert ex.path_stats()['hpf_chunked_total'] == ps['hpf_chunked_total'] and ex.seg_stats()['used']
    assert np.array_equal(r.clip_hashes(0), O.extract(short, O.Params())[1])
    # the boundary check forced to fail: every unit of the batch is re-done by the sequential kernel, same rows
    ex.set_pipeline(compact=0, seg=1, hpf_force_fail=True)
    try:
        r = ex.extract(clips=clips, want_hashes=True, want_peaks=True)
        st = ex.seg_stats()
        assert st['used'] and st['failed_units'] == len(clips), st
        for i, (pls, hs) in enumerate(want):
            assert np.array_equal(r.unit_peaks(i), pls[0]) and np.array_equal(r.clip_hashes(i), hs), i
    finally:
        ex.set_pipeline()
