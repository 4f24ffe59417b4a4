"""A fresh clone has no binary (the .so is git-ignored): prove that a CLEAN, forced compile of every template
instantiation for gfx950 still succeeds and exports the whole C ABI -- independent of whatever prebuilt
bayesianinference_amd/lib/libgphip.so travelled with the tree."""
import ctypes
import os

from bayesianinference_amd import _lib, build


def test_forced_clean_build_exports_every_declared_symbol(tmp_path):
    out = str(tmp_path / "libgphip_clean.so")
    path = build.build(force=True, out=out)
    assert path == out and os.path.getsize(out) > 500_000          # ~1.3 MB: kernels for fp64 + fp32, every role
    _lib.load()                                                    # map the HIP runtime the usual way first
    lib = ctypes.CDLL(out)
    missing = [s for s in _lib.declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.gphip_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.gphip_version()
    # the code object really targets gfx950 (no other architecture, no host fallback)
    blob = open(out, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob and b"gfx942" not in blob and b"gfx90a" not in blob
