"""Worker of tests/test_gpu_multiprocess.py::test_rank_local_failure_*: ONE rank of a multi-process job in which ONE rank is made
to fail locally in the middle of a collective sequence (fault injection options of the library).  What must hold: every rank
RETURNS from the call (nobody is left blocked in a collective), every rank reports an error for it, and the handles are usable
for the next call."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn  # noqa: E402


def attempt(fn):
    try:
        return {"ok": True, "value": fn()}
    except _lib.GphipError as exc:
        return {"ok": False, "status": exc.status, "msg": str(exc)}


def main():
    rank, world, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    n, d, kernel, panel = int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7])
    fault, fault_rank, fault_arg = sys.argv[8], int(sys.argv[9]), int(sys.argv[10])
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    h = _lib.Handle(X, y, kernel, device=0, rank=rank, world=world, comm_id=bytes(_lib.COMM_ID_BYTES))
    h.set_option("shard_min_n", 0)
    h.set_option("panel", panel)
    res = {"rank": rank}
    if fault == "alloc":
        # the advisor's case: dense workspace per rank, no in-place reads, and the FIRST sharded evaluation of the handle --
        # the failing rank has no layout at all when it has to keep the broadcasts going
        h.set_option("replicate_factor", 1)
        h.set_option("share_local_panels", 0)
        if rank == fault_rank:
            h.set_option("debug_fail_alloc", fault_arg)
        res["faulted"] = attempt(lambda: list(h.loglik_parts(th)))
    elif fault == "hip":
        res["warm"] = attempt(lambda: list(h.loglik_parts(th)))
        if rank == fault_rank:
            h.set_option("debug_fail_hip", fault_arg)
        res["faulted"] = attempt(lambda: list(h.loglik_parts(th)))
        h.set_option("debug_fail_hip", 0)
    elif fault == "predict":
        res["fit"] = attempt(lambda: h.fit(th))
        Xs = syn.make_test_points(32, d)
        if rank == fault_rank:
            h.set_option("panel", panel + 1)            # this rank's distributed factor no longer matches: a rank-LOCAL refusal
        res["faulted"] = attempt(lambda: [v.tolist() for v in h.predict(Xs)])
        h.set_option("panel", panel)
    res["after"] = attempt(lambda: list(h.loglik_parts(th)))      # the job carries on: same collective, everybody healthy
    h.close()
    with open(out_path, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
