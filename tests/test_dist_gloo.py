"""World-size-2 gloo test of the theta-sharded path (no GPU): each rank evaluates its own shard
with an injected evaluator (the CPU oracle as the checker) and the host-side merge reproduces the
single-process sweep exactly.  This is the N>1 logic of bench.py / nested sampling."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from bayesianinference_amd import distributed as D, synthetic as syn
    from oracle import gp_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y = syn.make_dataset(96, 2)                       # every rank regenerates the same data
    thetas = syn.theta_batch(7, "se_ard", 2)             # ragged: 7 units over 2 ranks
    calls = []

    def evaluate(th):
        calls.append(len(th))
        out = [orc.log_likelihood("se_ard", t, X, y, parts=True) for t in th]
        return np.array([o[0] for o in out]), np.array([o[3] for o in out])

    vals, info = D.sharded_map(evaluate, thetas, dist)
    # test-point sharding of the prediction: 11 points over 2 ranks
    Xs = syn.make_test_points(11, 2)
    npred = []

    def predict(P):
        npred.append(len(P))
        mu, sd = orc.predict_internal("se_ard", thetas[0], X, y, P)
        return mu, sd ** 2

    pm, pv = D.sharded_predict(predict, Xs, dist)
    q.put((rank, vals, info, sum(calls), pm, pv, sum(npred)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_theta_sharded_sweep_world2():
    from bayesianinference_amd import synthetic as syn
    from oracle import gp_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, y = syn.make_dataset(96, 2)
    thetas = syn.theta_batch(7, "se_ard", 2)
    want = np.array([orc.log_likelihood("se_ard", t, X, y) for t in thetas])
    by_rank = {r[0]: r for r in res}
    for r in (0, 1):
        np.testing.assert_array_equal(by_rank[r][1], want)          # identical merged result on every rank
        assert not by_rank[r][2].any()
    assert by_rank[0][3] == 4 and by_rank[1][3] == 3                 # 7 units dealt 4 + 3, no overlap
    mu, sd = orc.predict_internal("se_ard", thetas[0], X, y, syn.make_test_points(11, 2))
    for r in (0, 1):
        np.testing.assert_allclose(by_rank[r][4], mu, rtol=1e-8, atol=1e-11)   # LAPACK rounding differs with nrhs
        np.testing.assert_allclose(by_rank[r][5], sd ** 2, rtol=1e-8)
    assert by_rank[0][6] == 6 and by_rank[1][6] == 5                 # 11 test points dealt 6 + 5
