"""Covariance functions given as source text (gphip_create_custom): what can be checked WITHOUT a GPU.
(1) The oracle's function-valued kernel path against its own named kernels (so that the GPU parity tests of
    tests/test_gpu_custom_kernel.py compare against something pinned).
(2) The [rtc-begin] .. [rtc-end] region of csrc/gp_kernels.h -- the text the library compiles at run time around the caller's
    function -- still compiles on its own under hiprtc for gfx950, in both arithmetic types, with loop-style and CForm-style
    bodies; and a broken body is reported with the compiler's log.  hiprtc needs no GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from bayesianinference_amd import _lib, synthetic as syn
from oracle import gp_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SE_ARD_BODY = "T s = 0; for (int k = 0; k < D; ++k) { const T u = (X(k) - Y(k)) / P(k); s += u * u; } return P(D) * P(D) * exp((T)-0.5 * s);"


def se_ard_fn(A, B, p):
    d = A.shape[-1]
    return p[d] ** 2 * np.exp(-0.5 * (((A - B) / p[:d]) ** 2).sum(-1))


def test_oracle_function_valued_kernel_equals_named_kernel():
    X, y = syn.make_dataset(150, 3)
    Xs = syn.make_test_points(20, 3)
    th = syn.default_theta("se_ard", 3)
    ck = _lib.CustomKernel(SE_ARD_BODY, 4, fn=se_ard_fn)
    assert orc.is_custom(ck) and not orc.is_custom("se_ard") and orc.n_params(ck, 3, "const") == 6
    a = orc.log_likelihood("se_ard", th, X, y, parts=True)
    b = orc.log_likelihood(ck, th, X, y, parts=True)
    np.testing.assert_allclose(b[:3], a[:3], rtol=1e-12)
    ma, sa = orc.predict_internal("se_ard", th, X, y, Xs)
    mb, sb = orc.predict_internal(ck, th, X, y, Xs)
    np.testing.assert_allclose(mb, ma, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sb, sa, rtol=1e-10)
    thc = np.append(th, 0.3)
    assert abs(orc.log_likelihood(ck, thc, X, y, "const") - orc.log_likelihood("se_ard", thc, X, y, "const")) < 1e-9


def _hiprtc():
    for name in (os.environ.get("GPHIP_HIPRTC_PATH"), "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"):
        if not name:
            continue
        try:
            return C.CDLL(name)
        except OSError:
            continue
    return None


def _compile(rtc, body, ty):
    src = open(os.path.join(ROOT, "bayesianinference_amd", "csrc", "gp_kernels.h")).read()
    region = src[src.index("// [rtc-begin]"):src.index("// [rtc-end]")]
    full = ("#define GP_CUSTOM_KERNEL 1\n" + region + "\nnamespace gphip {\n"
            "template <typename A, typename B> __device__ auto Power(A a, B b) -> decltype(a * 1.0f) { return pow(a, (decltype(a * 1.0f))b); }\n"
            "template <typename A> __device__ A Exp(A a) { return exp(a); }\n"
            "template <typename T> __device__ T gphip_custom_k(PointRef<T> X, PointRef<T> Y, const double* __restrict__ Pp, int D) {\n"
            "#define P(k) ((T)Pp[(k)])\n" + body + "\n#undef P\n}\n}\n")
    prog = C.c_void_p()
    assert rtc.hiprtcCreateProgram(C.byref(prog), full.encode(), b"t.hip", 0, None, None) == 0
    names = [f"gphip::kbuild_kernel<{ty}, 0, 3>".encode(), f"gphip::custom_diag_kernel<{ty}>".encode(),
             f"gphip::custom_prep_kernel<{ty}>".encode()]
    for n in names:
        rtc.hiprtcAddNameExpression(prog, n)
    opts = (C.c_char_p * 3)(b"--offload-arch=gfx950", b"-O3", b"-std=c++17")
    rc = rtc.hiprtcCompileProgram(prog, 3, opts)
    n = C.c_size_t()
    rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
    log = C.create_string_buffer(max(n.value, 1))
    rtc.hiprtcGetProgramLog(prog, log)
    size = C.c_size_t(0)
    if rc == 0:
        rtc.hiprtcGetCodeSize(prog, C.byref(size))
        for nm in names:
            low = C.c_char_p()
            assert rtc.hiprtcGetLoweredName(prog, nm, C.byref(low)) == 0 and low.value
    rtc.hiprtcDestroyProgram(C.byref(prog))
    return rc, log.value.decode(errors="replace"), size.value


@pytest.mark.parametrize("ty", ["double", "float"])
def test_kernel_build_region_compiles_on_its_own_under_hiprtc(ty):
    rtc = _hiprtc()
    if rtc is None:
        pytest.skip("no libhiprtc.so on this machine")
    for body in (SE_ARD_BODY, "return Power(P(1),2)*Exp(-0.5*Power((X(0)-Y(0))/P(0),2));"):
        rc, log, size = _compile(rtc, body, ty)
        assert rc == 0 and size > 1000, log
    rc, log, _ = _compile(rtc, "return P(0) * undeclared_symbol;", ty)
    assert rc != 0 and "undeclared_symbol" in log


def test_rtc_region_has_no_host_only_dependencies():
    src = open(os.path.join(ROOT, "bayesianinference_amd", "csrc", "gp_kernels.h")).read()
    region = src[src.index("// [rtc-begin]"):src.index("// [rtc-end]")]
    body = "\n".join(ln for ln in region.splitlines() if not ln.lstrip().startswith("//"))
    assert "#include" not in body and "std::" not in body
