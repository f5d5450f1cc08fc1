"""CPU: the C-ABI library loads without a GPU and exports every symbol include/afp.h declares;
host-side constants equal the reference's numpy expressions; calls that need a GPU fail loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from audfprint_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from audfprint_amd import build
        build.build(verbose=False)
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'afp.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(afp_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'afp_status', 'afp_params', 'afp_handle'}
    assert len(declared) >= 20
    from audfprint_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_the_dynamic_symbol_table_is_exactly_the_header(lib):
    """the library is sealed (-fvisibility=hidden, AFP_API on the header's declarations, a linker version
    script): `nm -D` lists the functions of include/afp.h and NOTHING else -- no afp_launch_* launcher, no kernel handle."""
    import subprocess
    from audfprint_amd import _lib
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH], text=True)
    exported = set(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert exported == set(_lib.EXPORTS), (sorted(exported - set(_lib.EXPORTS))[:10], sorted(set(_lib.EXPORTS) - exported)[:10])
    hdr = open(os.path.join(ROOT, 'include', 'afp.h')).read()
    assert hdr.count('\nAFP_API ') == len(_lib.EXPORTS)            # every declaration carries the attribute


def test_abi_version_and_strerror(lib):
    from audfprint_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'afp.h')).read()
    assert int(re.search(r'#define AFP_ABI_VERSION (\d+)', hdr).group(1)) == _lib.AFP_ABI_VERSION == lib.afp_abi_version() == 2


def test_a_library_of_another_abi_version_is_refused(lib, monkeypatch):
    """version 2 widened the arrays afp_get_path_stats / afp_get_seg_stats write; a binding built for another
    version must not get as far as calling them."""
    from audfprint_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'AFP_ABI_VERSION', 1)
    with pytest.raises(_lib.AfpError, match='ABI version'):
        _lib.load()
    assert lib.afp_strerror(0) == b'ok'
    assert b'gfx950' in lib.afp_strerror(-6)
    for i in range(12):
        assert isinstance(lib.afp_kernel_name(i), bytes) and len(lib.afp_kernel_name(i)) > 0


def test_integration_md_names_every_export():
    """INTEGRATION.md §2 is the maintainer's map of the boundary -- every symbol of include/afp.h appears in it
    by its full name, and it names nothing the header does not declare."""
    import re
    from audfprint_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, 'INTEGRATION.md')).read()
    sec = doc[doc.index('## 2. The C ABI'):doc.index('ctypes stub (what')]
    named = set(re.findall(r'`(afp_[a-z0-9_]+)`', sec))
    assert named == set(_lib.EXPORTS), (sorted(set(_lib.EXPORTS) - named), sorted(named - set(_lib.EXPORTS)))


def test_param_struct_layout_matches_header():
    from audfprint_amd import _lib
    # double,double, 6 x int32, 16 x int32, 2 pointers
    assert ctypes.sizeof(_lib.AfpParams) == 8 + 8 + 6 * 4 + 16 * 4 + 2 * ctypes.sizeof(ctypes.c_void_p)


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from audfprint_amd import _lib
    from audfprint_amd.batch import Extractor
    assert lib.afp_device_count() == 0
    h = ctypes.c_void_p()
    assert lib.afp_create(0, ctypes.byref(h)) == -6          # AFP_ERR_NODEVICE
    with pytest.raises(_lib.AfpError):
        Extractor(0)
    from audfprint_amd.audfprint_analyze import Analyzer
    with pytest.raises(_lib.AfpError):
        Analyzer().find_peaks(np.zeros(4000, np.float32), 11025)


def test_missing_library_is_a_loud_error(monkeypatch):
    from audfprint_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libafp_hip.so')
    with pytest.raises(_lib.AfpError, match='no CPU fallback'):
        _lib.load()


def test_host_constants_equal_reference_expressions():
    from audfprint_amd.batch import host_constants
    from oracle import afp_oracle as O
    for density in (20.0, 70.0, 200.0):
        for shifts in (None, 0, 1, 2, 3, 4, 7):
            a_dec, window, gauss, offs = host_constants(density, 512, 256, 30.0, shifts)
            assert a_dec == float(O.a_dec_of(density))
            assert np.array_equal(window, O.hann_window())
            assert np.array_equal(gauss, O.gauss_table(256, 30.0)[256:512])
            assert offs == O.shift_offsets(shifts)
    G = O.gauss_table(256, 17.5)
    assert np.array_equal(G[256 - np.arange(256)], G[256 + np.arange(256)])     # symmetric: one half suffices


def test_unsupported_geometry_is_rejected():
    from audfprint_amd.batch import Extractor
    with pytest.raises(ValueError):
        Extractor.set_params(Extractor.__new__(Extractor), n_fft=1024)


def test_runtime_is_configured_explicitly_not_at_import():
    """VERDICT r4 #9 / ADVICE r4: importing the package leaves the environment alone; configure_runtime() sets
    GPU_MAX_HW_QUEUES=12 (a value of the user's wins), reports what it did, and WARNS when the HIP runtime of the process is
    already up (here: torch.cuda pretending to be initialised) -- the setting then cannot take effect and nothing is changed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}

    def run(code, **extra):
        return subprocess.check_output([sys.executable, '-W', 'always', '-c', code], cwd=root, env=dict(env, **extra), text=True,
                                       stderr=subprocess.STDOUT).strip()
    assert run("import os, audfprint_amd, audfprint_amd._lib, audfprint_amd.batch; print(os.environ.get('GPU_MAX_HW_QUEUES'))") == 'None'
    out = run("import os, audfprint_amd; i = audfprint_amd.configure_runtime(); print(os.environ['GPU_MAX_HW_QUEUES'], i['applied'], i['GPU_MAX_HW_QUEUES'])")
    assert out == '12 True 12'
    out = run("import os, audfprint_amd; i = audfprint_amd.configure_runtime(); print(os.environ['GPU_MAX_HW_QUEUES'], i['applied'])", GPU_MAX_HW_QUEUES='6')
    assert out == '6 False'
    code = ("import os, torch, audfprint_amd\n"
            "torch.cuda.is_initialized = lambda: True\n"
            "i = audfprint_amd.configure_runtime()\n"
            "print('ENV', os.environ.get('GPU_MAX_HW_QUEUES'), i['applied'], i['hip_was_initialised_before_configure'])\n"
            "audfprint_amd.configure_runtime()\n")
    out = run(code)
    assert 'ENV None False True' in out and out.count('RuntimeWarning') == 1 and 'could not be applied' in out, out
    # the first Extractor of a process configures the runtime itself (no GPU here: it raises right after)
    code = ("import os\nfrom audfprint_amd.batch import Extractor\n"
            "try:\n    Extractor(0)\nexcept Exception as e:\n    print(type(e).__name__)\n"
            "print(os.environ.get('GPU_MAX_HW_QUEUES'))\n")
    import torch
    if not torch.cuda.is_available():
        assert run(code).split() == ['AfpError', '12']
