"""CPU test: libsconf_hip.so builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports exactly the
entry points include/sconf.h declares.  No compute call is made."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'sconf.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sconf_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from lcasr_amd.hip import _lib
    assert os.path.exists(_lib.LIB_PATH)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 27
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/sconf.h but not exported'
    bound = set(_lib.PROTOTYPES) | set(_lib.PLAIN)
    assert bound == set(names), (bound ^ set(names))
    lib.sconf_version.restype = ctypes.c_int
    assert lib.sconf_version() >= 100
    lib.sconf_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.sconf_last_error(), bytes)


def test_host_side_argument_validation_sets_error():
    """Entry points validate shapes on the host before any launch (no GPU needed for the failure path)."""
    from lcasr_amd.hip import _lib
    lib = _lib.load()
    rc = lib.sconf_gemm_bf16(7, None, None, None, 8, 8, 8, 8, 8, 8, None, None, 8, None, 8, None, 8, 1.0, 0, 0, 1, None)
    assert rc != 0 and b'bad layout' in lib.sconf_last_error()
    rc = lib.sconf_norm_fwd(0, None, 0, None, None, None, 0, None, None, 4, 4096, 1e-5, None)
    assert rc != 0 and b'2048' in lib.sconf_last_error()
