"""Static guards on the HIP host code for bug classes that tests hit only by luck.

1. Null-stream memory operations.  Every handle works on its own NON-BLOCKING streams, which are not ordered with the
   null stream: a `hipMemset(...)` / `hipMemcpy(...)` of device memory that kernels on those streams read next is a
   race (round 2: the flag / ticket clears after a slot re-allocation -- wrong likelihoods and memory faults once per
   few thousand handles, found by scripts/gpu_api_fuzz.py).  Device memory is cleared / copied with the *Async forms
   on the handle's stream; the only allowed synchronous calls are the blocking host-to-device uploads of the exp table and of
   the inputs' mid-range vector in create_ctx, before any kernel of the handle exists.
2. No CUDA compatibility layer, no multi-backend dispatch (the build is gfx950-only by contract)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayesianinference_amd", "csrc")


def _code(path):
    text = open(path).read()
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def test_no_null_stream_memory_operations_on_device_buffers():
    allowed = {"hipMemcpy(h->dExp2, tab.data(), tab.size() * 8, hipMemcpyHostToDevice)",
               "hipMemcpy(h->dCentre, h->x_centre.data(), (size_t)d * 8, hipMemcpyHostToDevice)"}
    bad = []
    for name in ("gphip.hip", "gphip_multi.inc", "gp_kernels.h"):
        code = _code(os.path.join(CSRC, name))
        for m in re.finditer(r"\bhip(Memset|Memcpy|Memcpy2D|MemsetD8|MemsetD32)\s*\(([^;]*)\)\s*[;)]", code):
            call = re.sub(r"\s+", " ", m.group(0)).rstrip(";)").rstrip() + ")"
            call = call if call.count("(") == call.count(")") else call[:-1]
            if not any(call.startswith(a[:40]) for a in allowed):
                bad.append((name, call[:100]))
    assert not bad, bad


def test_gfx950_only_no_compat_layers():
    for name in os.listdir(CSRC):
        code = _code(os.path.join(CSRC, name))
        for token in ("__HIP_PLATFORM_AMD__", "__CUDACC__", "cuda_runtime", "cudaMalloc", "hipify", "__HIP_PLATFORM_NVIDIA__"):
            assert token not in code, (name, token)
