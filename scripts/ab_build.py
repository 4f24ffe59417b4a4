"""Developer A/B: builds variants of libgphip.so with different -D switches into bayesianinference_amd/lib/variants/
(in-tree so they travel to the GPU box; *.so is git-ignored).  Select one at run time with GPHIP_LIB=<path>.
   python scripts/ab_build.py name1:-DGP_TILE_PAD=512 name2:-DGP_TILE_PAD=0,-DGP_KB_INTERLEAVE=0 ..."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bayesianinference_amd", "csrc", "gphip.hip")
OUT = os.path.join(ROOT, "bayesianinference_amd", "lib", "variants")
os.makedirs(OUT, exist_ok=True)
def one(spec):
    name, _, flags = spec.partition(":")
    src = SRC
    if flags.startswith("src="):
        src, flags = flags[4:], ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.dirname(SRC), "-o",
           os.path.join(OUT, f"libgphip_{name}.so"), src, "-ldl", "-lpthread"] + [f for f in flags.split(",") if f]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr[-500:]
with ThreadPoolExecutor(4) as ex:
    for name, rc, err in ex.map(one, sys.argv[1:]):
        print(name, "ok" if rc == 0 else "FAILED\n" + err)
