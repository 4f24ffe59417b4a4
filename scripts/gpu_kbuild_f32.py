"""The fp32 Matern-5/2 kernel build of cfg 5 alone (N=65536, d=16: gphip_covariance is O(N^2) over PCIe, so the build is
driven through a likelihood evaluation at a size where it dominates less -- here: 8 evaluations at N=65536, of which
rocprofv3 reports the kbuild_kernel<float, 16, 1> launches separately)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 16
X, y = syn.make_dataset(n, d)
th = syn.default_theta("matern52_ard", d, dtype="f32")
h = _lib.Handle(X, y, "matern52_ard", dtype=32)
for i in range(3):
    print(h.loglik(th * (1 + 0.01 * i)))
h.close()
