"""The C-ABI library loads and exports every symbol include/vc_hip.h declares (no compute
calls, no GPU needed), and the ctypes table in _vc.py covers exactly those symbols."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'vc_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vc_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    import _vc
    if not os.path.exists(_vc.LIB_PATH):
        pytest.fail('libvc_hip.so not built (run __graft_entry__.build())')
    names = _declared_symbols()
    assert len(names) >= 10
    h = ctypes.CDLL(_vc.LIB_PATH)
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, missing
    assert set(_vc._SIGS) == set(names), set(_vc._SIGS) ^ set(names)


def test_version_and_errors_without_gpu():
    import _vc
    lib = _vc.lib()
    assert lib.vc_version() == 1
    assert lib.vc_target_arch() == b'gfx950'
    # argument validation happens before any HIP call
    cfg = _vc.FrontendCfg(16000, 80, 400, 200, 80, 40, 0.97, 0.003, 0.01, 0.01, 0.01, 1, 1, 1)
    rc = lib.vc_frontend_host_tables(ctypes.byref(cfg), None, None)
    assert rc == 1 and b'n_fft' in lib.vc_last_error()
    with pytest.raises(_vc.VCError):
        _vc.check(rc)


def test_struct_layout_matches_header():
    import _vc
    assert ctypes.sizeof(_vc.FrontendCfg) == 14 * 4
