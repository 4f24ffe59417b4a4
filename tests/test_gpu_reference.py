"""HIP path against outputs of the REAL reference (tests/golden/reference_<case>.json, produced by
oracle/make_reference_golden.wl on a machine with a Wolfram kernel).  Skipped -- with the reason "parity
unpinned" -- while no such file is committed; the tolerance is BASELINE.json's: 1e-8 relative."""
import glob
import json
import os

import numpy as np
import pytest

from bayesianinference_amd import _lib

pytestmark = pytest.mark.gpu


def test_hip_matches_reference_fixture(golden_dir):
    files = sorted(f for f in glob.glob(os.path.join(golden_dir, "reference_*.json"))
                   if not f.endswith("reference_inputs.json"))
    if not files:
        pytest.skip("parity unpinned: no reference-produced fixture present (oracle/make_reference_golden.wl)")
    inputs = {c["name"]: c for c in json.load(open(os.path.join(golden_dir, "reference_inputs.json")))["cases"]}
    for f in files:
        ref = json.load(open(f))
        c = inputs[ref["name"]]
        X, y = np.array(c["X"]), np.array(c["y"])
        n = len(y)
        h = _lib.Handle(X, y, c["kernel"], c["mean"])
        for i, th in enumerate(np.array(ref["thetas"])):
            ll, ld, qd, info = h.loglik_parts(th)
            if ref["loglik_is_sentinel"][i]:
                assert info != 0, (ref["name"], i)         # the shim substitutes $MachineLogZero
                continue
            assert info == 0
            assert abs(ll - ref["loglik"][i]) <= 1e-8 * max(abs(ref["loglik"][i]), n), (ref["name"], i)
        pts = np.array(ref["pred_points"])
        for i, (mu, sd) in enumerate(zip(ref["pred_mu"], ref["pred_sd"])):
            assert h.fit(np.array(ref["thetas"])[i]) == 0
            m, v = h.predict(pts)
            np.testing.assert_allclose(m, mu, rtol=1e-7, atol=1e-8)
            np.testing.assert_allclose(np.sqrt(v), sd, rtol=1e-7)
        h.close()
