"""ctypes binding of the C ABI in include/gphip.h (the drop-in boundary).

The library is the product: there is no CPU fallback.  If `libgphip.so` is missing or no gfx950
device is visible, every compute entry point raises `GphipError` loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libgphip.so")
if os.environ.get("GPHIP_LIB"):        # developer A/B runs: another BUILD of the same HIP library (scripts/ab_build.py)
    LIB_PATH = os.environ["GPHIP_LIB"]
HEADER = os.path.join(os.path.dirname(PKG), "include", "gphip.h")

OK = 0
INFO_OK, INFO_NOT_SPD, INFO_NAN = 0, 1, 2
KERNEL_IDS = {"se": 0, "se_ard": 1, "matern52": 2, "matern52_ard": 3, "null": 4, "matern32": 5, "matern32_ard": 6,
              "rq": 7, "rq_ard": 8}
_OPS = {"+": 1, "*": 2}


def kernel_id(name: str) -> int:
    """Named kernel or a composed form -> the integer of include/gphip.h.  Grammar (spaces ignored):
        term [(+|*) term] [+const]      term in se, se_ard, matern52, matern52_ard, matern32, matern32_ard, rq, rq_ard
    e.g. "se+const" (the reference's own example, BGP:16), "se_ard+matern32", "rq*se_ard+const".  theta layout:
    [term 1: l.., (alpha), sf] [term 2: l.., (alpha), sf] [c] sn [mu]."""
    key = name.replace(" ", "").lower()
    if key in KERNEL_IDS:
        return KERNEL_IDS[key]
    offset = 0
    if key.endswith("+const"):
        key, offset = key[:-len("+const")], 1
    for sym, op in _OPS.items():
        if sym in key:
            a, b = key.split(sym, 1)
            break
    else:
        a, b, op = key, None, 0
    if a not in KERNEL_IDS or a == "null" or (b is not None and (b not in KERNEL_IDS or b == "null")):
        raise GphipError(1, f"unknown kernel {name!r}")
    return KERNEL_IDS[a] | ((KERNEL_IDS[b] if b else 0) << 8) | (op << 16) | (offset << 20) | (1 << 24)

MEAN_IDS = {"zero": 0, "const": 1}
PROFILE_CLASSES = ("kbuild", "potrf", "trsm", "gemm_panel", "syrk_trailing", "eval_total", "predict_epilogue")
COMM_ID_BYTES = 128

_STATUS = {1: "bad argument", 2: "dimension mismatch", 3: "HIP runtime failure",
           4: "handle not fitted", 5: "no gfx950 device", 6: "unsupported"}


class GphipError(RuntimeError):
    def __init__(self, status: int, msg: str = ""):
        self.status = status
        super().__init__(f"gphip status {status} ({_STATUS.get(status, '?')}): {msg}")


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_h = C.c_void_p

_SIGNATURES = {
    "gphip_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                               _ip, C.c_int, C.POINTER(_h)]),
    "gphip_create_custom": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(_h)]),
    "gphip_create_custom_devices": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int, C.c_int, C.c_int, _ip, C.c_int,
                                              C.POINTER(_h)]),
    "gphip_create_custom_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p, C.POINTER(_h)]),
    "gphip_create_error": (C.c_char_p, []),
    "gphip_custom_compile": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, _ip]),
    "gphip_custom_compile_d": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, _ip]),
    "gphip_kernel_parse": (C.c_int, [C.c_char_p, C.c_int64, _ip]),
    "gphip_cform_to_body": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int64]),
    "gphip_tab_prior_sample": (C.c_int, [_dp, _dp, C.c_int, C.c_int64, C.c_int, C.c_uint64, _dp]),
    "gphip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "gphip_create_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(_h)]),
    "gphip_comm_info": (C.c_int, [_h, _ip, _ip, C.POINTER(C.c_char_p)]),
    "gphip_destroy": (C.c_int, [_h]),
    "gphip_num_params": (C.c_int, [_h, _ip]),
    "gphip_loglik": (C.c_int, [_h, _dp, C.c_int, _dp, _ip]),
    "gphip_loglik_batch": (C.c_int, [_h, _dp, C.c_int, C.c_int, _dp, _ip]),
    "gphip_loglik_parts": (C.c_int, [_h, _dp, C.c_int, _dp, _dp, _ip]),
    "gphip_loglik_grad": (C.c_int, [_h, _dp, C.c_int, _dp, _dp, _ip]),
    "gphip_fit": (C.c_int, [_h, _dp, C.c_int, _ip]),
    "gphip_predict": (C.c_int, [_h, C.c_void_p, C.c_int64, _dp, _dp]),
    "gphip_predict_samples": (C.c_int, [_h, _dp, C.c_int, C.c_int, C.c_void_p, C.c_int64, _dp, _dp, _ip]),
    "gphip_loglik_batch_pw": (C.c_int, [_h, _dp, C.c_int, C.c_int, _dp, _dp, _dp, _ip]),
    "gphip_fit_pw": (C.c_int, [_h, _dp, C.c_int, _dp, _dp, _ip]),
    "gphip_predict_pw": (C.c_int, [_h, C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp]),
    "gphip_predict_samples_pw": (C.c_int, [_h, _dp, C.c_int, C.c_int, _dp, _dp, C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp, _ip]),
    "gphip_covariance": (C.c_int, [_h, _dp, C.c_int, _dp]),
    "gphip_covariance_batch": (C.c_int, [_h, _dp, C.c_int, C.c_int, _dp]),
    "gphip_cross_covariance": (C.c_int, [_h, _dp, C.c_int, C.c_void_p, C.c_int64, _dp, _dp]),
    "gphip_solve": (C.c_int, [_h, _dp, C.c_int64, _dp]),
    "gphip_logdet": (C.c_int, [_h, _dp]),
    "gphip_set_option": (C.c_int, [_h, C.c_char_p, C.c_double]),
    "gphip_get_option": (C.c_int, [_h, C.c_char_p, C.POINTER(C.c_double)]),
    "gphip_get_profile": (C.c_int, [_h, C.c_int, _dp, _dp, _dp, _dp]),
    "gphip_reset_profile": (C.c_int, [_h]),
    "gphip_sync": (C.c_int, [_h]),
    "gphip_ns_default_options": (C.c_int, [C.c_void_p]),
    "gphip_nested_sampling": (C.c_int, [_h, _dp, _ip, C.c_void_p, C.c_void_p, C.c_void_p, _dp, C.c_int64, _dp, _dp, _dp, _dp,
                                        C.POINTER(C.c_int64), _dp, C.POINTER(C.c_int64)]),
    "gphip_ns_crude_weights": (C.c_int, [_dp, _dp, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), _dp, _dp, _dp]),
    "gphip_factor_bytes": (C.c_int, [_h, C.c_int, _dp]),
    "gphip_set_streams": (C.c_int, [_h, C.c_void_p, C.c_void_p]),
    "gphip_dist_num_panels": (C.c_int, [_h, _ip]),
    "gphip_dist_panel_shape": (C.c_int, [_h, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gphip_dist_begin": (C.c_int, [_h, _dp, C.c_int, C.c_int, C.c_int]),
    "gphip_dist_factor_panel": (C.c_int, [_h, C.c_int, C.c_void_p]),
    "gphip_dist_update": (C.c_int, [_h, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "gphip_dist_end": (C.c_int, [_h, _dp, _dp, _ip]),
    "gphip_last_error": (C.c_char_p, [_h]),
    "gphip_version": (C.c_char_p, []),
    "gphip_device_count": (C.c_int, [_ip]),
}

_lib = None


def declared_symbols() -> list[str]:
    """Every function name include/gphip.h declares (used by the symbol-export test)."""
    with open(HEADER) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gphip_[a-z_]+)\s*\(", text)))


def load():
    """dlopen the in-tree library (built by `python -m bayesianinference_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GphipError(3, f"{LIB_PATH} is missing -- run `python -m bayesianinference_amd.build` "
                            "(hipcc, gfx950).  There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  A process must end up with
    # ONE HIP runtime, or torch later reports "No HIP GPUs are available" and torch streams cannot be
    # handed to this library (dist_cholesky.py).  Importing torch first makes the dynamic loader
    # resolve libgphip's libamdhip64.so.7 dependency to the copy torch already mapped.
    if os.environ.get("GPHIP_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        if os.environ.get("GPHIP_LIB") and not hasattr(lib, name):
            continue                                   # developer A/B against an OLDER build of the library
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def device_count() -> int:
    n = C.c_int(0)
    load().gphip_device_count(C.byref(n))
    return n.value


def comm_unique_id() -> bytes:
    """RCCL unique id (rank 0 creates it, the host distributes it to every rank: gphip_create_rank)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().gphip_comm_unique_id(buf)
    if rc != OK:
        raise GphipError(rc, "gphip_comm_unique_id: no RCCL could be bound (set GPHIP_RCCL_PATH)")
    return buf.raw


def _d(a: np.ndarray):
    return a.ctypes.data_as(_dp)


class NsOptions(C.Structure):
    """gphip_ns_options (include/gphip.h)"""
    _fields_ = [("pool", C.c_int), ("max_iterations", C.c_int), ("min_iterations", C.c_int), ("mc_steps", C.c_int),
                ("walkers", C.c_int), ("termination_fraction", C.c_double), ("min_accept", C.c_double),
                ("max_accept", C.c_double), ("seed", C.c_uint64)]


LOGPRIOR_FN = C.CFUNCTYPE(C.c_double, _dp, C.c_int, C.c_void_p)


def ns_crude_weights(points, loglik, pool: int):
    """gphip_ns_crude_weights: (order, logX, crude log weights, log evidence) -- calculateWeightsCrude, BS:818-835."""
    pts = np.ascontiguousarray(np.atleast_2d(np.asarray(points, dtype=np.float64)))
    ll = np.ascontiguousarray(np.asarray(loglik, dtype=np.float64))
    m, p = pts.shape
    order = np.zeros(m, dtype=np.int64)
    logx, logw, z = np.zeros(m), np.zeros(m), C.c_double(0.0)
    rc = load().gphip_ns_crude_weights(_d(pts), _d(ll), m, p, int(pool), order.ctypes.data_as(C.POINTER(C.c_int64)), _d(logx),
                                       _d(logw), C.byref(z))
    if rc != OK:
        raise GphipError(rc, "gphip_ns_crude_weights")
    return order, logx, logw, z.value


class CustomKernel:
    """ANY covariance function k(x, x'; p) for the device (the reference takes any `kernel @@ points[[{i,j}]]`, BGP:29-33):
    `body` = C++ statements in X(k), Y(k) (coordinates of the two points), P(k) (hyper-parameters), D (dimension), T (the
    arithmetic type) that `return` the covariance without the nugget -- see gphip_create_custom in include/gphip.h;
    theta of such a handle = [p_0 .. p_{nparams-1}, sn (, mu)].  `fn(A, B, p)` (optional) = the same function in numpy on
    broadcastable point arrays [.., d] -- what the CPU oracle evaluates in the tests."""

    def __init__(self, body: str, nparams: int, fn=None, name: str = "custom"):
        self.body, self.nparams, self.fn, self.name = str(body), int(nparams), fn, name

    def __repr__(self):
        return f"CustomKernel({self.name!r}, nparams={self.nparams})"


class Handle:
    """Owns one gphip_handle: training data resident on the device (gphip_create .. gphip_destroy)."""

    def __init__(self, X, y, kernel="se_ard", mean: str = "zero", dtype: int = 64, device=None,
                 rank=None, world=None, comm_id: bytes | None = None):
        """device: None (current device), one ordinal, or a LIST of ordinals = one multi-device handle in this
        process (a repeated ordinal = virtual ranks sharing a GPU).  rank/world/comm_id: this process is one
        rank of a multi-process job (gphip_create_rank; comm_id from comm_unique_id() of rank 0).
        kernel: a name of the grammar of kernel_id(), or a CustomKernel (ANY covariance function, given as the source text of
        its body: compiled at run time into the library's kernel build, gphip_create_custom)."""
        lib = load()
        if isinstance(kernel, CustomKernel):
            self._init_custom(lib, X, y, kernel, mean, dtype, device, rank, world, comm_id)
            return
        X = np.ascontiguousarray(np.atleast_2d(np.asarray(X, dtype=np.float64)))
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).ravel())
        if X.shape[0] != y.shape[0]:
            raise GphipError(2, "Input and output data are not of same length")     # BGP:251-253
        if mean not in MEAN_IDS:
            raise GphipError(1, f"unknown mean {mean!r}")
        kid = kernel_id(kernel)
        self.N, self.d = X.shape
        self.kernel, self.mean, self.dtype = kernel, mean, int(dtype)
        self._lib = lib
        self._h = _h()
        devs, nd = (None, 0)
        if device is not None:
            lst = [int(v) for v in device] if isinstance(device, (list, tuple)) else [int(device)]
            devs, nd = (C.c_int * len(lst))(*lst), len(lst)
        if comm_id is not None:
            if rank is None or world is None or len(comm_id) != COMM_ID_BYTES:
                raise GphipError(1, "rank, world and a 128-byte comm_id go together")
            rc = lib.gphip_create_rank(X.ctypes.data, y.ctypes.data, self.N, self.d, kid, MEAN_IDS[mean],
                                       dtype, -1 if device is None else int(devs[0]), int(rank), int(world),
                                       C.c_char_p(comm_id), C.byref(self._h))
        else:
            rc = lib.gphip_create(X.ctypes.data, y.ctypes.data, self.N, self.d, kid,
                                  MEAN_IDS[mean], dtype, devs, nd, C.byref(self._h))
        if rc != OK:
            self._h = None
            why = {5: "no gfx950 GPU visible, or a device ordinal that does not exist",
                   6: "unsupported combination (null kernel on a multi-device handle, d > 32, dtype other than 64 / 32, or no "
                      "RCCL could be bound for a multi-process handle: set GPHIP_RCCL_PATH)",
                   1: "bad argument (unknown kernel / mean id, rank outside the world)",
                   2: "bad shape (N < 1 or d < 1)"}.get(rc, "is a gfx950 GPU visible?")
            raise GphipError(rc, "gphip_create failed: " + why)
        p = C.c_int(0)
        lib.gphip_num_params(self._h, C.byref(p))
        self.p = p.value

    def _init_custom(self, lib, X, y, kernel, mean, dtype, device, rank=None, world=None, comm_id=None):
        X = np.ascontiguousarray(np.atleast_2d(np.asarray(X, dtype=np.float64)))
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).ravel())
        if X.shape[0] != y.shape[0]:
            raise GphipError(2, "Input and output data are not of same length")     # BGP:251-253
        if mean not in MEAN_IDS:
            raise GphipError(1, f"unknown mean {mean!r}")
        self.N, self.d = X.shape
        self.kernel, self.mean, self.dtype = kernel, mean, int(dtype)
        self._lib = lib
        self._h = _h()
        lst = [] if device is None else ([int(v) for v in device] if isinstance(device, (list, tuple)) else [int(device)])
        devs = (C.c_int * max(len(lst), 1))(*lst) if lst else None
        if comm_id is not None:
            if rank is None or world is None or len(comm_id) != COMM_ID_BYTES:
                raise GphipError(1, "rank, world and a 128-byte comm_id go together")
            rc = lib.gphip_create_custom_rank(X.ctypes.data, y.ctypes.data, self.N, self.d, kernel.body.encode(), int(kernel.nparams),
                                              MEAN_IDS[mean], dtype, lst[0] if lst else -1, int(rank), int(world),
                                              C.c_char_p(comm_id), C.byref(self._h))
        else:
            rc = lib.gphip_create_custom_devices(X.ctypes.data, y.ctypes.data, self.N, self.d, kernel.body.encode(), int(kernel.nparams),
                                                 MEAN_IDS[mean], dtype, devs, len(lst), C.byref(self._h))
        if rc != OK:
            self._h = None
            raise GphipError(rc, "gphip_create_custom failed: " + (lib.gphip_create_error() or b"").decode())
        p = C.c_int(0)
        lib.gphip_num_params(self._h, C.byref(p))
        self.p = p.value

    # -- helpers -------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != OK:
            raise GphipError(rc, (self._lib.gphip_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gphip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: float):
        self._check(self._lib.gphip_set_option(self._h, name.encode(), float(value)))

    def get_option(self, name: str) -> float:
        v = C.c_double()
        self._check(self._lib.gphip_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    # -- hot path ------------------------------------------------------------------------
    def loglik(self, theta):
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        out, info = C.c_double(0.0), C.c_int(0)
        self._check(self._lib.gphip_loglik(self._h, _d(th), th.size, C.byref(out), C.byref(info)))
        return out.value, info.value

    def loglik_parts(self, theta):
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        out, info = C.c_double(0.0), C.c_int(0)
        parts = np.zeros(2)
        self._check(self._lib.gphip_loglik_parts(self._h, _d(th), th.size, C.byref(out), _d(parts),
                                                 C.byref(info)))
        return out.value, parts[0], parts[1], info.value

    def loglik_batch(self, Theta):
        Th = np.ascontiguousarray(np.atleast_2d(np.asarray(Theta, dtype=np.float64)))
        B, p = Th.shape
        out = np.zeros(B)
        info = np.zeros(B, dtype=np.int32)
        self._check(self._lib.gphip_loglik_batch(self._h, _d(Th), B, p, _d(out), info.ctypes.data_as(_ip)))
        return out, info

    @staticmethod
    def _rows(a, rows: int, cols: int, what: str):
        """optional per-point array -> (contiguous float64 [rows, cols] or None, ctypes pointer or None)"""
        if a is None:
            return None, None
        arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(rows, -1))
        if arr.shape != (rows, cols):
            raise GphipError(2, f"{what} must have shape ({rows}, {cols})")
        return arr, _d(arr)

    def loglik_batch_pw(self, Theta, mean_train=None, nugget_train=None):
        """gphip_loglik_batch_pw: point-dependent mean m(x_i) / nugget nu(x_i) VALUES per theta, each [B, N] or None
        (BGP:37, 300: the host evaluates the functions, the device gets the values)."""
        Th = np.ascontiguousarray(np.atleast_2d(np.asarray(Theta, dtype=np.float64)))
        B, p = Th.shape
        mt, mtp = self._rows(mean_train, B, self.N, "mean_train")
        nt, ntp = self._rows(nugget_train, B, self.N, "nugget_train")
        out = np.zeros(B)
        info = np.zeros(B, dtype=np.int32)
        self._check(self._lib.gphip_loglik_batch_pw(self._h, _d(Th), B, p, mtp, ntp, _d(out), info.ctypes.data_as(_ip)))
        return out, info

    def fit_pw(self, theta, mean_train=None, nugget_train=None) -> int:
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        mt, mtp = self._rows(mean_train, 1, self.N, "mean_train")
        nt, ntp = self._rows(nugget_train, 1, self.N, "nugget_train")
        info = C.c_int(0)
        self._check(self._lib.gphip_fit_pw(self._h, _d(th), th.size, mtp, ntp, C.byref(info)))
        return info.value

    def predict_pw(self, Xs, mean_test=None, nugget_test=None):
        Xs = np.ascontiguousarray(np.atleast_2d(np.asarray(Xs, dtype=np.float64)))
        if Xs.shape[1] != self.d:
            raise GphipError(2, "test points have the wrong dimension")
        M = Xs.shape[0]
        ms, msp = self._rows(mean_test, 1, M, "mean_test")
        ns, nsp = self._rows(nugget_test, 1, M, "nugget_test")
        mean, var = np.zeros(M), np.zeros(M)
        self._check(self._lib.gphip_predict_pw(self._h, Xs.ctypes.data, M, msp, nsp, _d(mean), _d(var)))
        return mean, var

    def predict_samples_pw(self, Thetas, Xs, mean_train=None, nugget_train=None, mean_test=None, nugget_test=None):
        Th = np.ascontiguousarray(np.atleast_2d(np.asarray(Thetas, dtype=np.float64)))
        Xs = np.ascontiguousarray(np.atleast_2d(np.asarray(Xs, dtype=np.float64)))
        if Xs.shape[1] != self.d:
            raise GphipError(2, "test points have the wrong dimension")
        S, M = Th.shape[0], Xs.shape[0]
        mt, mtp = self._rows(mean_train, S, self.N, "mean_train")
        nt, ntp = self._rows(nugget_train, S, self.N, "nugget_train")
        ms, msp = self._rows(mean_test, S, M, "mean_test")
        ns, nsp = self._rows(nugget_test, S, M, "nugget_test")
        mean, var = np.zeros((S, M)), np.zeros((S, M))
        info = np.zeros(S, dtype=np.int32)
        self._check(self._lib.gphip_predict_samples_pw(self._h, _d(Th), S, Th.shape[1], mtp, ntp, Xs.ctypes.data, M, msp, nsp,
                                                       _d(mean), _d(var), info.ctypes.data_as(_ip)))
        return mean, var, info

    def loglik_grad(self, theta):
        """(loglik, grad[p], info)."""
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        out, info = C.c_double(0.0), C.c_int(0)
        grad = np.zeros(th.size)
        self._check(self._lib.gphip_loglik_grad(self._h, _d(th), th.size, C.byref(out), _d(grad), C.byref(info)))
        return out.value, grad, info.value

    def fit(self, theta) -> int:
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        info = C.c_int(0)
        self._check(self._lib.gphip_fit(self._h, _d(th), th.size, C.byref(info)))
        return info.value

    def predict(self, Xs):
        Xs = np.ascontiguousarray(np.atleast_2d(np.asarray(Xs, dtype=np.float64)))
        if Xs.shape[1] != self.d:
            raise GphipError(2, "test points have the wrong dimension")
        M = Xs.shape[0]
        mean, var = np.zeros(M), np.zeros(M)
        self._check(self._lib.gphip_predict(self._h, Xs.ctypes.data, M, _d(mean), _d(var)))
        return mean, var

    def predict_samples(self, Thetas, Xs):
        """All posterior samples in one batched pass: (mean[S,M], var[S,M], info[S])."""
        Th = np.ascontiguousarray(np.atleast_2d(np.asarray(Thetas, dtype=np.float64)))
        Xs = np.ascontiguousarray(np.atleast_2d(np.asarray(Xs, dtype=np.float64)))
        if Xs.shape[1] != self.d:
            raise GphipError(2, "test points have the wrong dimension")
        S, M = Th.shape[0], Xs.shape[0]
        mean, var = np.zeros((S, M)), np.zeros((S, M))
        info = np.zeros(S, dtype=np.int32)
        self._check(self._lib.gphip_predict_samples(self._h, _d(Th), S, Th.shape[1], Xs.ctypes.data, M,
                                                    _d(mean), _d(var), info.ctypes.data_as(_ip)))
        return mean, var, info

    def covariance(self, theta):
        """theta[p] -> K[N, N]; Theta[B, p] -> K[B, N, N] (the Listable form, BGP:59)."""
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64))
        if th.ndim == 2:
            K = np.zeros((th.shape[0], self.N, self.N))
            self._check(self._lib.gphip_covariance_batch(self._h, _d(th), th.shape[0], th.shape[1], _d(K)))
            return K
        th = th.ravel()
        K = np.zeros((self.N, self.N))
        self._check(self._lib.gphip_covariance(self._h, _d(th), th.size, _d(K)))
        return K

    def cross_covariance(self, theta, Xs):
        """compiledKandKappa (BGP:91-124): (k[N, M], kappa[M])."""
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        Xs = np.ascontiguousarray(np.atleast_2d(np.asarray(Xs, dtype=np.float64)))
        if Xs.shape[1] != self.d:
            raise GphipError(2, "test points have the wrong dimension")
        M = Xs.shape[0]
        k, kappa = np.zeros((self.N, M)), np.zeros(M)
        self._check(self._lib.gphip_cross_covariance(self._h, _d(th), th.size, Xs.ctypes.data, M, _d(k), _d(kappa)))
        return k, kappa

    def comm_info(self) -> dict:
        w, nl, name = C.c_int(0), C.c_int(0), C.c_char_p()
        self._check(self._lib.gphip_comm_info(self._h, C.byref(w), C.byref(nl), C.byref(name)))
        return {"world": w.value, "local": nl.value, "comm": (name.value or b"").decode()}

    def solve(self, rhs):
        rhs = np.asarray(rhs, dtype=np.float64)
        vec = rhs.ndim == 1
        B = np.ascontiguousarray(rhs.reshape(self.N, -1).T)      # each rhs contiguous
        out = np.zeros_like(B)
        self._check(self._lib.gphip_solve(self._h, _d(B), B.shape[0], _d(out)))
        return out[0] if vec else out.T.copy()

    def logdet(self) -> float:
        out = C.c_double(0.0)
        self._check(self._lib.gphip_logdet(self._h, C.byref(out)))
        return out.value

    def nested_sampling(self, box, prior_kind=None, logprior=None, start=None, cap=None, **options):
        """gphip_nested_sampling: the native batched sampler.  options: pool, max_iterations, min_iterations, mc_steps,
        walkers, termination_fraction, min_accept, max_accept, seed.  Returns a dict with Points, LogLikelihood,
        LogPriorPDF, AcceptanceRate (generation order), CrudeLogEvidence, LikelihoodEvaluations, SamplePoolSize."""
        o = NsOptions()
        self._check(self._lib.gphip_ns_default_options(C.byref(o)))
        for k, v in options.items():
            if not hasattr(o, k):
                raise GphipError(1, f"unknown sampler option {k!r}")
            setattr(o, k, v)
        box = np.ascontiguousarray(np.asarray(box, dtype=np.float64).reshape(self.p, 2))
        kinds = None if prior_kind is None else np.ascontiguousarray(np.asarray(prior_kind, dtype=np.int32))
        cb = None
        if logprior is not None:
            cb = LOGPRIOR_FN(lambda th, p, _u: float(logprior(np.ctypeslib.as_array(th, shape=(p,)).copy())))
        st = None if start is None else np.ascontiguousarray(np.asarray(start, dtype=np.float64).reshape(o.pool, self.p))
        cap = int(cap or (o.pool + max(o.max_iterations, o.min_iterations) + 1))
        pts, ll, lp, ar = np.zeros((cap, self.p)), np.zeros(cap), np.zeros(cap), np.zeros(cap)
        ns, ne, z = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        self._check(self._lib.gphip_nested_sampling(
            self._h, _d(box), None if kinds is None else kinds.ctypes.data_as(_ip), C.cast(cb, C.c_void_p) if cb else None, None,
            C.byref(o), None if st is None else _d(st), cap, _d(pts), _d(ll), _d(lp), _d(ar), C.byref(ns), C.byref(z), C.byref(ne)))
        m = ns.value
        return {"Points": pts[:m].copy(), "LogLikelihood": ll[:m].copy(), "LogPriorPDF": lp[:m].copy(),
                "AcceptanceRate": ar[:m].copy(), "SamplePoolSize": int(o.pool), "GeneratedNestedSamples": m - int(o.pool),
                "TotalSamples": m, "CrudeLogEvidence": z.value, "LikelihoodEvaluations": ne.value, "Seed": int(o.seed)}

    # -- measurement ---------------------------------------------------------------------
    def reset_profile(self):
        self._check(self._lib.gphip_reset_profile(self._h))

    def profile(self) -> dict:
        res = {}
        for i, name in enumerate(PROFILE_CLASSES):
            v = [C.c_double(0.0) for _ in range(4)]
            self._check(self._lib.gphip_get_profile(self._h, i, *[C.byref(x) for x in v]))
            res[name] = dict(ms=v[0].value, launches=v[1].value, flops=v[2].value, bytes=v[3].value)
        return res

    def sync(self):
        self._check(self._lib.gphip_sync(self._h))

    def factor_bytes(self, member: int = 0) -> float:
        """device bytes local rank `member` holds for factor storage right now (workspace + own panels + receive buffers)"""
        v = C.c_double(0.0)
        self._check(self._lib.gphip_factor_bytes(self._h, member, C.byref(v)))
        return v.value

    # -- multi-GPU block-cyclic Cholesky steps (driven by dist_cholesky.py) ---------------
    def set_streams(self, main_stream: int, panel_stream: int):
        self._check(self._lib.gphip_set_streams(self._h, C.c_void_p(main_stream), C.c_void_p(panel_stream)))

    def dist_num_panels(self) -> int:
        n = C.c_int(0)
        self._check(self._lib.gphip_dist_num_panels(self._h, C.byref(n)))
        return n.value

    def dist_panel_shape(self, k: int):
        r, c = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.gphip_dist_panel_shape(self._h, k, C.byref(r), C.byref(c)))
        return r.value, c.value

    def dist_begin(self, theta, rank: int, world: int):
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).ravel())
        self._check(self._lib.gphip_dist_begin(self._h, _d(th), th.size, rank, world))

    @staticmethod
    def _dev_ptr(packed) -> int:
        """device address of a packed-panel buffer: a torch tensor (data_ptr) or a raw integer"""
        return int(packed.data_ptr()) if hasattr(packed, "data_ptr") else int(packed)

    def dist_factor_panel(self, k: int, packed):
        self._check(self._lib.gphip_dist_factor_panel(self._h, k, C.c_void_p(self._dev_ptr(packed))))

    def dist_update(self, k: int, packed, j_first: int, j_last: int, on_panel_stream: bool):
        self._check(self._lib.gphip_dist_update(self._h, k, C.c_void_p(self._dev_ptr(packed)), j_first, j_last,
                                                int(on_panel_stream)))

    def dist_end(self):
        ld, qd, info = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        self._check(self._lib.gphip_dist_end(self._h, C.byref(ld), C.byref(qd), C.byref(info)))
        return ld.value, qd.value, info.value
