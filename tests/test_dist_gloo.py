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


# ---------------------------------------------------------------------------------------------
# The 1-D block-cyclic distributed Cholesky schedule (dist_cholesky.DistributedCholesky) under REAL
# torch.distributed broadcasts (gloo, world_size 2): ownership, look-ahead order, packed-panel
# rotation, corner tile, scalar all-reduce.  The per-rank compute steps come from a numpy stand-in for
# the gphip_dist_* C-ABI calls (test infrastructure, same contract), so no GPU is needed.
# ---------------------------------------------------------------------------------------------
class NumpyPanelBackend:
    TB = 128

    def __init__(self, K_full, r_vec, panel):
        import numpy as np
        self.np = np
        n = K_full.shape[0]
        self.N, self.dtype, self.panel = n, 64, panel
        self.npad = (n + 127) // 128 * 128
        self.nt = self.npad // 128
        self.Kfull, self.r = K_full, r_vec

    def set_streams(self, a, b):
        pass

    def dist_num_panels(self):
        return (self.nt + self.panel - 1) // self.panel

    def dist_panel_shape(self, k):
        k0 = k * self.panel
        k1 = min(k0 + self.panel, self.nt)
        return (self.nt + 1 - k0) * 128, (k1 - k0) * 128

    def _owned(self, j):
        return (j // (self.panel * 128)) % self.world == self.rank

    def dist_begin(self, theta, rank, world):
        np = self.np
        self.rank, self.world = rank, world
        m = self.npad + 128
        A = np.full((m, m), np.nan)                       # NaN everywhere this rank must never read
        Kp = np.eye(self.npad)
        Kp[:self.N, :self.N] = self.Kfull
        for j in range(self.npad):
            if self._owned(j):
                A[j:self.npad, j] = Kp[j:, j]
                A[self.npad:, j] = 0.0
                A[self.npad, j] = self.r[j] if j < self.N else 0.0
        if rank == 0:
            A[self.npad:, self.npad:] = 0.0
        self.A, self.logdet = A, 0.0

    def dist_factor_panel(self, k, packed):
        np = self.np
        rows, cols = self.dist_panel_shape(k)
        c0 = k * self.panel * 128
        P = self.A[c0:, c0:c0 + cols]
        L11 = np.linalg.cholesky(np.tril(P[:cols]) + np.tril(P[:cols], -1).T)
        P[:cols] = L11
        P[cols:] = np.linalg.solve(L11, P[cols:].T).T
        self.logdet += 2.0 * np.log(np.diag(L11)).sum()
        packed.numpy().reshape(cols, rows).T[...] = P      # column-major rows x cols

    def dist_update(self, k, packed, j_first, j_last, on_panel_stream):
        rows, cols = self.dist_panel_shape(k)
        P = packed.numpy().reshape(cols, rows).T
        k0t = k * self.panel
        nouter = self.dist_num_panels()
        for j in range(max(j_first, k + 1), min(j_last, nouter + 1)):
            if j == nouter:
                if self.rank != 0:
                    continue
                c_lo, c_hi = self.nt * 128, (self.nt + 1) * 128
            else:
                if j % self.world != self.rank:
                    continue
                c_lo, c_hi = j * self.panel * 128, min((j + 1) * self.panel, self.nt) * 128
            off = c_lo - k0t * 128
            self.A[c_lo:, c_lo:c_hi] -= P[off:] @ P[off:off + (c_hi - c_lo)].T

    def dist_end(self):
        quad = -self.A[self.npad, self.npad] if self.rank == 0 else 0.0
        return self.logdet, quad, 0


def _chol_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from bayesianinference_amd import synthetic as syn
    from bayesianinference_amd.dist_cholesky import DistributedCholesky, TorchDistComm
    from oracle import gp_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y = syn.make_dataset(900, 3)                      # 8 tiles: 4 outer panels of 2, last one ragged
    th = syn.default_theta("se_ard", 3)
    K = orc.covariance_matrix("se_ard", th, X)
    be = NumpyPanelBackend(K, y, panel=2)
    dc = DistributedCholesky({rank: be}, TorchDistComm(dist), device="cpu")
    out = [dc.loglik(th) for _ in range(2)]               # second pass reuses the rotating buffers
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_distributed_cholesky_schedule_world2():
    from bayesianinference_amd import synthetic as syn
    from oracle import gp_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_chol_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=200) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, y = syn.make_dataset(900, 3)
    th = syn.default_theta("se_ard", 3)
    want = orc.log_likelihood("se_ard", th, X, y, parts=True)
    for r in (0, 1):
        for ll, ld, qd, info in res[r]:
            assert info == 0
            assert ld == pytest.approx(want[1], rel=1e-11) and qd == pytest.approx(want[2], rel=1e-10)
            assert ll == pytest.approx(want[0], rel=1e-11)
