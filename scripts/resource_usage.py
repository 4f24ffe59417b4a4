"""Register / scratch / LDS use of every kernel in libgphip (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
   python scripts/resource_usage.py [source.hip]      (CPU only: hipcc cross-compiles)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "bayesianinference_amd", "csrc", "gphip.hip")
with tempfile.TemporaryDirectory() as td:
    res = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "bayesianinference_amd", "csrc"),
                          "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(td, "x.so"), src, "-ldl", "-lpthread"],
                         capture_output=True, text=True)
txt = res.stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].split(" [")[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
def g(b, k):
    m = re.search(k + r": (\d+)", b)
    return m.group(1) if m else "?"
for b, dn in zip(blocks, dem):
    dn = dn.replace("gphip::", "")[:100]
    print("%-100s V=%s A=%s spill=%s scratch=%s occ=%s lds=%s" % (dn, g(b, "VGPRs"), g(b, "AGPRs"), g(b, "VGPRs Spill"),
          g(b, r"ScratchSize \[bytes/lane\]"), g(b, r"Occupancy \[waves/SIMD\]"), g(b, r"LDS Size \[bytes/block\]")))
