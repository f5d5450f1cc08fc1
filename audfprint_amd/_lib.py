"""ctypes binding of libafp_hip.so (the C ABI in include/afp.h).

The product has NO CPU fallback: if the HIP library is missing or no GPU is usable the
functions here raise, loudly.  Build the library with ``python -m audfprint_amd.build``
(or ``__graft_entry__.build()``)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and a stream's wait on an event
# is a barrier packet in ITS hardware queue: every packet queued behind it -- other streams' kernels and copies that share
# the queue -- waits too.  A pipelined ingest here runs ~10 streams (one per context, three stage streams, the table stream,
# the upload stream): in the r04 trace of the 12 500-clip job the fetch of batch 1 sat 9 ms behind "wait for the upload of
# batch 3" in a queue the two contexts' streams shared, and the PCIe link idled.  With a queue per stream the false
# dependencies go (job 67 -> 62 ms, C3 unchanged).  The runtime reads the variable when it initialises (the first HIP call
# of the process).  Importing this package does NOT touch the environment (it is a drop-in module inside somebody else's
# process): configure_runtime() below does, explicitly -- bench.py calls it first thing, the first Extractor of a process
# calls it, and it WARNS when the runtime is already up and the setting can no longer take effect.
DEFAULT_HW_QUEUES = 12
_runtime = dict(configured=False, requested=None, applied=False, reason=None, hip_was_up=None)


def _hip_already_initialised():
    """Best effort, without making a HIP call: has anything in this process brought the runtime up?  (This library's
    own first call, or torch.cuda -- is_available() / device_count() count: they call hipGetDeviceCount.)"""
    import sys
    if _lib is not None and _runtime.get('touched'):
        return True
    try:        # the HSA runtime under HIP opens /dev/kfd when it initialises: an open descriptor = the flags have been read
        for fd in os.listdir('/proc/self/fd'):
            try:
                if os.readlink('/proc/self/fd/' + fd) == '/dev/kfd':
                    return True
            except OSError:
                pass
    except OSError:
        pass
    t = sys.modules.get('torch')
    if t is not None:
        try:
            if t.cuda.is_initialized() or getattr(t.cuda, '_cached_device_count', None) is not None:
                return True
        except Exception:
            pass
    return False


def configure_runtime(hw_queues=DEFAULT_HW_QUEUES, quiet=False):
    """Ask the HIP runtime for `hw_queues` hardware queues (GPU_MAX_HW_QUEUES) -- BEFORE the process's first HIP call, which
    is when the runtime reads it.  A value the user exported wins.  If the runtime is already initialised (e.g. `import torch;
    torch.cuda.init()` came first) nothing can be changed any more: a RuntimeWarning says so (a pipelined ingest then shares
    4 hardware queues between ~10 streams: the 12 500-clip job measured 67 ms instead of 62; single batches are unaffected).
    Idempotent; returns runtime_info()."""
    import warnings
    if not _runtime['configured']:
        _runtime['configured'] = True
        _runtime['requested'] = int(hw_queues)
        up = _hip_already_initialised()
        _runtime['hip_was_up'] = bool(up)
        if 'GPU_MAX_HW_QUEUES' in os.environ:
            _runtime['reason'] = 'GPU_MAX_HW_QUEUES=%s was set by the user' % os.environ['GPU_MAX_HW_QUEUES'] + \
                                 (' (the runtime was already initialised when audfprint_amd looked)' if up else '')
        elif up:
            _runtime['reason'] = 'the HIP runtime was initialised before audfprint_amd.configure_runtime() ran'
            if not quiet:
                warnings.warn('audfprint_amd: GPU_MAX_HW_QUEUES=%d could not be applied -- the HIP runtime of this process is '
                              'already initialised (call audfprint_amd.configure_runtime() or export the variable before the '
                              'first HIP / torch.cuda call).  Pipelined ingests will share the default 4 hardware queues.'
                              % int(hw_queues), RuntimeWarning, stacklevel=2)
        else:
            os.environ['GPU_MAX_HW_QUEUES'] = str(int(hw_queues))
            _runtime['applied'] = True
            _runtime['reason'] = 'set by audfprint_amd.configure_runtime() before the first HIP call'
    return runtime_info(query=False)


def runtime_info(query=True):
    """What the library runs on: GPU_MAX_HW_QUEUES as the runtime will read / has read it, who set it, and -- with query=True,
    which makes a HIP call -- the HIP version of the build vs the runtime actually bound (a PyTorch wheel's bundled runtime or
    the system one) and where that runtime was mapped from."""
    info = dict(GPU_MAX_HW_QUEUES=os.environ.get('GPU_MAX_HW_QUEUES'), requested=_runtime['requested'], applied=_runtime['applied'],
                reason=_runtime['reason'], hip_was_initialised_before_configure=_runtime['hip_was_up'])
    if query:
        lib = load()
        out = (C.c_int32 * 4)()
        if lib.afp_runtime_info(out) == 0:
            _runtime['touched'] = True
            def ver(v):
                return '%d.%d.%d' % (v // 10000000, (v // 100000) % 100, v % 100000)
            info.update(hip_build_version=ver(out[0]), hip_runtime_version=ver(out[1]), hip_driver_version=ver(out[2]), devices=int(out[3]),
                        hip_versions_match=bool(out[0] // 100000 == out[1] // 100000))
        try:
            with open('/proc/self/maps') as f:
                libs = sorted(set(ln.split()[-1] for ln in f if 'libamdhip64' in ln))
            info['hip_runtime_path'] = libs
        except OSError:
            pass
    return info


LIB_PATH = os.environ.get('AFP_LIB_PATH') or os.path.join(HERE, 'lib', 'libafp_hip.so')   # (override: A/B builds)

AFP_ABI_VERSION = 2           # include/afp.h (2: afp_get_path_stats / afp_get_seg_stats write eight int32)
AFP_MAX_SHIFTS = 16
AFP_MAX_PKS = 64
AFP_NKERNELS = 12
WANT_HASHES, WANT_PEAKS, KEEP_DEBUG, WANT_LANDMARKS = 1, 2, 4, 8
UNIT_EMPTY, UNIT_ZERO, UNIT_CORR, UNIT_TIE, UNIT_NONFINITE, UNIT_NEARTIE = 1, 2, 4, 8, 16, 32

# every symbol include/afp.h declares (tests/test_abi_cpu.py checks the library exports them)
EXPORTS = ['afp_abi_version', 'afp_build_id', 'afp_strerror', 'afp_last_hip_error', 'afp_device_count', 'afp_create',
           'afp_destroy', 'afp_set_stream', 'afp_set_params', 'afp_set_workspace_limit',
           'afp_workspace_bytes', 'afp_extract_device', 'afp_extract_host', 'afp_result_counts',
           'afp_fetch_hashes', 'afp_fetch_peaks', 'afp_fetch_unit_flags', 'afp_result_device_ptrs',
           'afp_get_seg_stats', 'afp_set_pipeline', 'afp_set_seg_force_fail', 'afp_set_compact_force_timeout', 'afp_get_path_stats', 'afp_fetch_all', 'afp_fetch_unit_tie_frames', 'afp_clock_probe_start', 'afp_clock_probe_stop', 'afp_set_timing', 'afp_reset_timings', 'afp_get_timings', 'afp_kernel_name', 'afp_debug_fetch',
           'afp_pairs_from_peaks', 'afp_fetch_landmarks', 'afp_hashes_from_landmarks', 'afp_prune_spectrogram',
           'afp_extract_device_s16', 'afp_extract_host_s16', 'afp_extract_device_f64', 'afp_extract_host_f64',
           'afp_table_create', 'afp_table_upload', 'afp_table_download', 'afp_table_store', 'afp_table_store_device', 'afp_table_replay_overflow', 'afp_mt_randint_replay', 'afp_table_fetch_overflow', 'afp_table_patch', 'afp_table_merge', 'afp_table_merge_device',
           'afp_table_fetch_merge_overflow', 'afp_table_device_ptrs', 'afp_table_clip_counts',
           'afp_table_get_hits', 'afp_table_fetch_hits', 'afp_set_stage_streams', 'afp_stream_create_cu_range', 'afp_stream_destroy',
           'afp_table_count_ids', 'afp_table_fetch_id_counts', 'afp_table_skew_hist', 'afp_table_fetch_skew_hist',
           'afp_table_select_hits', 'afp_table_fetch_selected', 'afp_table_hits_max_time',
           'afp_table_download_filled', 'afp_table_pack', 'afp_table_packed_device_ptrs', 'afp_table_fetch_packed',
           'afp_table_merge_packed', 'afp_table_merge_packed_device', 'afp_host_threads', 'afp_pinned_alloc', 'afp_pinned_free', 'afp_runtime_info', 'afp_set_neartie_eps', 'afp_host_prefault', 'afp_retired_bytes']


class AfpParams(C.Structure):
    _fields_ = [('a_dec', C.c_double), ('hpf_pole', C.c_double),
                ('maxpksperframe', C.c_int32), ('maxpairsperpeak', C.c_int32),
                ('targetdf', C.c_int32), ('mindt', C.c_int32), ('targetdt', C.c_int32),
                ('nshifts', C.c_int32), ('shift_offsets', C.c_int32 * AFP_MAX_SHIFTS),
                ('window', C.POINTER(C.c_double)), ('gauss', C.POINTER(C.c_double))]


class AfpError(RuntimeError):
    """A libafp_hip call failed.  `status` is the afp_status the library returned (0 when the error was raised by the binding
    itself); `refused` says the library turned the call down on its arguments / state before doing anything."""
    status = 0

    @property
    def refused(self):
        return self.status in (-1, -2, -5)        # AFP_ERR_ARG, AFP_ERR_PARAM, AFP_ERR_STATE


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so with the same SONAME as the system
    one.  Whichever is loaded first serves the whole process, and mixing them (ours first, torch
    later) leaves torch without a GPU.  So if torch is installed, map ITS runtime before
    libafp_hip.so binds (no `import torch` needed).  AFP_HIP_RUNTIME=system skips this."""
    import sys
    if 'torch' in sys.modules or os.environ.get('AFP_HIP_RUNTIME', '') == 'system':
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load libafp_hip.so once per process; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AfpError('audfprint_amd: %s is missing -- build it with `python -m audfprint_amd.build` '
                       '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
    P = C.POINTER
    lib.afp_abi_version.restype = C.c_int
    lib.afp_build_id.restype = C.c_char_p
    lib.afp_strerror.restype = C.c_char_p
    lib.afp_strerror.argtypes = [C.c_int]
    lib.afp_last_hip_error.restype = C.c_char_p
    lib.afp_device_count.restype = C.c_int
    lib.afp_create.argtypes = [C.c_int, P(vp)]
    lib.afp_destroy.argtypes = [vp]
    lib.afp_destroy.restype = None
    lib.afp_set_stream.argtypes = [vp, vp]
    lib.afp_set_stage_streams.argtypes = [vp, vp, vp, vp]
    lib.afp_stream_create_cu_range.argtypes = [C.c_int, C.c_int, C.c_int, P(vp)]
    lib.afp_stream_destroy.argtypes = [vp]
    lib.afp_set_params.argtypes = [vp, P(AfpParams)]
    lib.afp_set_workspace_limit.argtypes = [vp, i64]
    lib.afp_workspace_bytes.argtypes = [vp, P(i64), i32, u32]
    lib.afp_workspace_bytes.restype = i64
    lib.afp_extract_device.argtypes = [vp, vp, P(i64), i32, u32]
    # (buffers of the per-file hot path go in as plain addresses: c_void_p takes an int, a ctypes pointer or None alike, and
    #  `arr.ctypes.data` costs half of `arr.ctypes.data_as(...)`)
    lib.afp_extract_host.argtypes = [vp, vp, vp, i32, u32]
    lib.afp_extract_device_s16.argtypes = [vp, vp, P(i64), i32, u32]
    lib.afp_extract_host_s16.argtypes = [vp, vp, vp, i32, u32]
    lib.afp_extract_device_f64.argtypes = [vp, vp, P(i64), i32, u32]
    lib.afp_extract_host_f64.argtypes = [vp, vp, vp, i32, u32]
    lib.afp_pairs_from_peaks.argtypes = [vp, P(i32), P(i64), i32, u32]
    lib.afp_fetch_landmarks.argtypes = [vp, P(i32), P(i64), P(i64)]
    lib.afp_hashes_from_landmarks.argtypes = [vp, P(i32), i64, P(i32)]
    lib.afp_prune_spectrogram.argtypes = [vp, P(C.c_double), i32, C.c_double, P(C.c_uint8), P(C.c_uint8), P(C.c_uint8)]
    lib.afp_table_create.argtypes = [vp, i32, i32, i32]
    lib.afp_table_upload.argtypes = [vp, P(C.c_uint32), P(i32)]
    lib.afp_table_download.argtypes = [vp, P(C.c_uint32), P(i32)]
    lib.afp_table_store.argtypes = [vp, P(i32), P(i64), P(i32), i32, P(i64)]
    lib.afp_table_fetch_overflow.argtypes = [vp, P(i32)]
    lib.afp_table_store_device.argtypes = [vp, vp, vp, i64, P(i32), i32, P(i64)]
    lib.afp_table_replay_overflow.argtypes = [vp, P(C.c_uint32), P(i32), P(i64)]
    lib.afp_mt_randint_replay.argtypes = [P(C.c_uint32), P(i32), P(i32), i64, P(i32)]
    lib.afp_table_patch.argtypes = [vp, P(i32), i64]
    lib.afp_table_merge.argtypes = [vp, P(C.c_uint32), P(i32), i32, i32, P(i64)]
    lib.afp_table_merge_device.argtypes = [vp, vp, vp, i32, i32, P(i64)]
    lib.afp_table_fetch_merge_overflow.argtypes = [vp, P(i32), P(i32), P(C.c_uint32)]
    lib.afp_table_device_ptrs.argtypes = [vp, P(vp), P(vp)]
    lib.afp_table_download_filled.argtypes = [vp, P(C.c_uint32), P(i32), P(i64)]
    lib.afp_table_pack.argtypes = [vp, P(i64)]
    lib.afp_table_packed_device_ptrs.argtypes = [vp, P(vp), P(vp), P(i64)]
    lib.afp_table_fetch_packed.argtypes = [vp, P(C.c_uint32), P(i32)]
    lib.afp_table_merge_packed.argtypes = [vp, P(C.c_uint32), i64, P(i32), i32, i32, P(i64)]
    lib.afp_table_merge_packed_device.argtypes = [vp, vp, vp, i32, i32, P(i64)]
    lib.afp_host_threads.restype = C.c_int
    lib.afp_pinned_alloc.argtypes = [C.c_int, i64, P(vp)]
    lib.afp_pinned_free.argtypes = [vp]
    lib.afp_runtime_info.argtypes = [P(i32)]
    lib.afp_set_neartie_eps.argtypes = [vp, C.c_double]
    lib.afp_host_prefault.argtypes = [vp, i64]
    lib.afp_retired_bytes.restype = i64
    lib.afp_table_clip_counts.argtypes = [vp]
    lib.afp_table_get_hits.argtypes = [vp, P(i32), i64, P(i64)]
    lib.afp_table_fetch_hits.argtypes = [vp, P(i32)]
    lib.afp_table_count_ids.argtypes = [vp, P(i64)]
    lib.afp_table_fetch_id_counts.argtypes = [vp, P(i32), P(i32)]
    lib.afp_table_skew_hist.argtypes = [vp, P(i32), i32, P(i32), P(i32)]
    lib.afp_table_fetch_skew_hist.argtypes = [vp, P(i32)]
    lib.afp_table_select_hits.argtypes = [vp, P(i32), P(i32), P(i32), i32, P(i64)]
    lib.afp_table_fetch_selected.argtypes = [vp, P(i32), P(i64)]
    lib.afp_table_hits_max_time.argtypes = [vp, P(i32)]
    lib.afp_result_counts.argtypes = [vp, P(i64), P(i64), P(i64)]
    lib.afp_fetch_hashes.argtypes = [vp, P(i32), P(i64)]
    lib.afp_fetch_peaks.argtypes = [vp, P(i32), P(i64)]
    lib.afp_fetch_unit_flags.argtypes = [vp, P(i32)]
    lib.afp_result_device_ptrs.argtypes = [vp, P(vp), P(vp), P(vp), P(vp)]
    lib.afp_set_timing.argtypes = [vp, C.c_int]
    lib.afp_fetch_unit_tie_frames.argtypes = [vp, P(i32), P(i32)]
    lib.afp_fetch_all.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.afp_get_seg_stats.argtypes = [vp, P(i32)]
    lib.afp_set_pipeline.argtypes = [vp, i32, i32, i32, i32, i32, i32]
    lib.afp_set_seg_force_fail.argtypes = [vp, i32]
    lib.afp_set_compact_force_timeout.argtypes = [vp, i32]
    lib.afp_get_path_stats.argtypes = [vp, P(i32)]
    lib.afp_clock_probe_start.argtypes = [vp, C.c_int]
    lib.afp_clock_probe_stop.argtypes = [vp, P(C.c_double)]
    lib.afp_reset_timings.argtypes = [vp]
    lib.afp_get_timings.argtypes = [vp, P(C.c_double), P(i64)]
    lib.afp_kernel_name.argtypes = [C.c_int]
    lib.afp_kernel_name.restype = C.c_char_p
    lib.afp_debug_fetch.argtypes = [vp, C.c_int, vp, i64]
    lib.afp_debug_fetch.restype = i64
    if lib.afp_abi_version() != AFP_ABI_VERSION:
        raise AfpError('libafp_hip.so speaks ABI version %d, this binding version %d (include/afp.h: AFP_ABI_VERSION) -- rebuild '
                       'with `python -m audfprint_amd.build`' % (lib.afp_abi_version(), AFP_ABI_VERSION))
    # the .so is a git-ignored build product: refuse one that was not compiled from the sources in this tree
    from . import build as _build
    want = _build.source_id()
    got = lib.afp_build_id().decode()
    if want is not None and got != want and not os.environ.get('AFP_ALLOW_STALE_LIB'):
        raise AfpError('audfprint_amd: %s was built from other sources (build id %s, tree %s) -- rebuild with '
                       '`python -m audfprint_amd.build`' % (LIB_PATH, got, want))
    _lib = lib
    return lib


def check(status, what=''):
    if status < 0:
        lib = load()
        msg = lib.afp_strerror(int(status)).decode()
        if status == -3:
            msg += ': ' + lib.afp_last_hip_error().decode()
        e = AfpError('%s failed: %s' % (what or 'libafp_hip call', msg))
        e.status = int(status)
        raise e
    return status
