"""Worker of tests/test_gpu_multiprocess.py: ONE rank of a multi-process job (gphip_create_rank).  Runs torch-free
(GPHIP_NO_TORCH=1) so that the library binds the collective library named by $GPHIP_RCCL_PATH."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn  # noqa: E402


def main():
    rank, world, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    n, d, kernel, panel = int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7])
    X, y = syn.make_dataset(n, d)                        # every rank regenerates the same data
    th = syn.default_theta(kernel, d)
    h = _lib.Handle(X, y, kernel, device=0, rank=rank, world=world, comm_id=bytes(_lib.COMM_ID_BYTES))
    h.set_option("shard_min_n", 0)
    h.set_option("panel", panel)
    res = {"rank": rank, "comm": h.comm_info()}
    res["parts"] = list(h.loglik_parts(th))              # collective: same theta, same order on every rank
    res["parts2"] = list(h.loglik_parts(th * 1.05))
    bad = th.copy()
    bad[0] = np.nan
    res["nan"] = list(h.loglik(bad))
    res["fit"] = h.fit(th)                               # collective; every rank keeps its own panels of L
    Xs = syn.make_test_points(64, d)
    lo, hi = 64 * rank // world, 64 * (rank + 1) // world     # each rank predicts ITS shard: collective (the factor's panels
    mu, var = h.predict(Xs[lo:hi])                            # stream through every rank once more)
    res["bytes"] = h.factor_bytes()
    res["mu"], res["var"], res["shard"] = mu.tolist(), var.tolist(), [lo, hi]
    res["logdet"] = h.logdet()
    res["alpha_head"] = h.solve(y)[:5].tolist()
    h.close()
    with open(out_path, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
