"""Worker of tests/test_gpu_multiprocess.py::test_single_process_grouped_rccl_path: ONE process, several ranks, the
library's ncclCommInitAll + grouped-broadcast path, bound to the tests-only collective library (torch-free)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn  # noqa: E402

out_path, n, d, kernel, world, panel = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
X, y = syn.make_dataset(n, d)
th = syn.default_theta(kernel, d)
h = _lib.Handle(X, y, kernel, device=[0] * world)
h.set_option("shard_min_n", 0)
h.set_option("panel", panel)
res = {"comm": h.comm_info(), "parts": list(h.loglik_parts(th)), "again": list(h.loglik_parts(th))}
res["fit"] = h.fit(th)
Xs = syn.make_test_points(700, d)
mu, var = h.predict(Xs)                                   # sharded over the ranks (every rank holds the factor)
res["mu"], res["var"] = mu.tolist(), var.tolist()
h.close()
json.dump(res, open(out_path, "w"))
