"""Worker of tests/test_gpu_multiprocess.py::test_create_failure_on_one_rank_fails_every_rank: ONE rank of a multi-process job
whose covariance function (source text) does not compile on ONE rank.  gphip_create_custom_rank joins the communicator first
and all-reduces a create status, so every rank must RETURN from the create call with an error -- nobody is left in
ncclCommInitRank -- and a second, healthy create of the same job must work."""
import json
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn  # noqa: E402

GOOD = "T s = 0; for (int k = 0; k < D; ++k) { const T u = (X(k) - Y(k)) / P(k); s += u * u; } return P(D) * P(D) * exp((T)-0.5 * s);"
BAD = "return P(0) * this_symbol_does_not_exist;"


def main():
    rank, world, out_path, bad_rank = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    n, d = 700, 3
    X, y = syn.make_dataset(n, d)
    res = {"rank": rank}
    ids = [bytes(_lib.COMM_ID_BYTES), bytes([1]) + bytes(_lib.COMM_ID_BYTES - 1)]      # two communicators, one per attempt
    try:
        h = _lib.Handle(X, y, _lib.CustomKernel(BAD if rank == bad_rank else GOOD, d + 1), device=0, rank=rank, world=world, comm_id=ids[0])
        h.close()
        res["first"] = {"ok": True}
    except _lib.GphipError as exc:
        res["first"] = {"ok": False, "status": exc.status, "msg": str(exc)}
    h = _lib.Handle(X, y, _lib.CustomKernel(GOOD, d + 1), device=0, rank=rank, world=world, comm_id=ids[1])
    h.set_option("shard_min_n", 0)
    h.set_option("panel", 2)
    res["second"] = list(h.loglik_parts(syn.default_theta("se_ard", d)))
    h.close()
    with open(out_path, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
