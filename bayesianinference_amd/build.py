"""Builds the in-tree HIP shared library (gfx950 only) with hipcc.

    python -m bayesianinference_amd.build          # -> bayesianinference_amd/lib/libgphip.so
hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG, "csrc", "gphip.hip")
DEPS = [SRC, os.path.join(PKG, "csrc", "gp_kernels.h"), os.path.join(PKG, "csrc", "gp_trsv.h"), os.path.join(PKG, "csrc", "gphip_multi.inc"),
        os.path.join(PKG, "csrc", "gphip_sampler.inc"), os.path.join(PKG, "csrc", "rtc_dyn.h"), os.path.join(PKG, "csrc", "gp_dual.h"),
        os.path.join(PKG, "csrc", "gphip_hostlogic.inc"),
        os.path.join(PKG, "csrc", "rccl_dyn.h"), os.path.join(os.path.dirname(PKG), "include", "gphip.h")]
LIB = os.path.join(PKG, "lib", "libgphip.so")


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libgphip.so (ROCm >= 7.0 required)")


def rocm_prefix() -> str:
    """ROCm installation the found hipcc belongs to ($ROCM_PATH, else <hipcc>/../..), not a hard-coded /opt/rocm."""
    env = os.environ.get("ROCM_PATH")
    if env and os.path.isdir(env):
        return env
    return os.path.dirname(os.path.dirname(os.path.realpath(hipcc_path())))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in DEPS if os.path.exists(p))


def build(force: bool = False, verbose: bool = False, out: str | None = None) -> str:
    """out: build to this path instead of the in-tree library (always a full compile; tests/test_clean_build.py)."""
    if out is None and not force and not needs_build():
        return LIB
    target = out or LIB
    os.makedirs(os.path.dirname(target), exist_ok=True)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-I" + os.path.join(PKG, "csrc"),      # (-I: rtc_dyn.h embeds gp_kernels.h with .incbin)
           "-o", target, SRC, "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return target


ROOT = os.path.dirname(PKG)
WL_SHIM = os.path.join(PKG, "csrc", "librarylink_shim.cpp")
WL_STUB_DIR = os.path.join(ROOT, "tests", "wl_stub")
WL_STUB_LIB = os.path.join(WL_STUB_DIR, "libgphip_wl_stub.so")


def build_wl_stub(force: bool = False, verbose: bool = False) -> str:
    """TEST build of the LibraryLink shim: the production source csrc/librarylink_shim.cpp compiled with g++ against
    the tests-only stand-in tests/wl_stub/WolframLibrary.h, together with the fake WolframLibraryData driver, linked
    to the in-tree libgphip.so.  (A production build uses the real header of a Wolfram installation, INTEGRATION.md.)"""
    srcs = [WL_SHIM, os.path.join(WL_STUB_DIR, "shim_driver.cpp")]
    deps = srcs + [os.path.join(WL_STUB_DIR, "WolframLibrary.h"), os.path.join(ROOT, "include", "gphip.h"), LIB]   # (srcs[0] = librarylink_shim.cpp)
    if not force and os.path.exists(WL_STUB_LIB) and all(os.path.getmtime(p) <= os.path.getmtime(WL_STUB_LIB) for p in deps):
        return WL_STUB_LIB
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("g++ not found: cannot build the LibraryLink shim test library")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-I" + WL_STUB_DIR,
           "-I" + os.path.join(ROOT, "include"), "-o", WL_STUB_LIB] + srcs + \
          ["-L" + os.path.dirname(LIB), "-lgphip", "-Wl,-rpath,$ORIGIN/../../bayesianinference_amd/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed on the LibraryLink shim:\n" + res.stdout + res.stderr)
    return WL_STUB_LIB


FAKE_RCCL_SRC = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
FAKE_RCCL_LIB = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def build_fake_rccl(force: bool = False, verbose: bool = False) -> str:
    """TESTS-ONLY collective library over shared memory (tests/fake_rccl/fake_rccl.cpp): lets the multi-PROCESS path
    (gphip_create_rank) run with two ranks on a one-GPU box, where real RCCL refuses two ranks on one device."""
    if not force and os.path.exists(FAKE_RCCL_LIB) and os.path.getmtime(FAKE_RCCL_SRC) <= os.path.getmtime(FAKE_RCCL_LIB):
        return FAKE_RCCL_LIB
    rocm = rocm_prefix()
    cmd = [hipcc_path(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-x", "c++", "-D__HIP_PLATFORM_AMD__",
           "-I" + os.path.join(rocm, "include"), "-o", FAKE_RCCL_LIB, FAKE_RCCL_SRC, "-L" + os.path.join(rocm, "lib"),
           "-lamdhip64", "-lpthread", "-lrt"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building the fake collective library failed:\n" + res.stdout + res.stderr)
    return FAKE_RCCL_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_wl_stub(force="--force" in sys.argv, verbose=True))
    print(build_fake_rccl(force="--force" in sys.argv, verbose=True))
