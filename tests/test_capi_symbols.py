"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/gphip.h declares (no compute calls -- there is no GPU here)."""
import ctypes

import pytest

from bayesianinference_amd import _lib, build


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    names = _lib.declared_symbols()
    assert "gphip_loglik" in names and "gphip_predict" in names and len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/gphip.h but not exported"
        assert name in _lib._SIGNATURES, f"{name} has no ctypes signature"


def test_version_and_status_contract(lib):
    assert b"gfx950" in lib.gphip_version()
    n = ctypes.c_int(-1)
    assert lib.gphip_device_count(ctypes.byref(n)) == 0 and n.value >= 0
    # argument validation happens before any device work
    h = ctypes.c_void_p()
    assert lib.gphip_create(None, None, 4, 1, 0, 0, 64, None, 0, ctypes.byref(h)) == 1
    assert lib.gphip_destroy(None) == 0
    assert lib.gphip_last_error(None) == b"null handle"
