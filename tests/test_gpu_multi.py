"""Multi-device handles behind the C ABI (include/gphip.h: gphip_create with ndev > 1, gphip_create_rank),
SURVEY.md §8b/§8e.  A 1-GPU box exercises them by listing the same device ordinal several times: every
listed entry becomes a rank with its own context, workspace, three streams and packed-panel buffers; the
schedule, ownership (panel j -> rank j % world), packing, look-ahead ordering, unpack-on-receive replication of
L and the scalar reduction are exactly what runs on 8 GPUs -- only the transport differs (device copies for
virtual ranks; RCCL, bound with dlopen, once the ordinals are distinct).  The RCCL binding itself is exercised
at world size 1 through gphip_create_rank (ncclCommInitRank / ncclBroadcast / ncclAllReduce on one GPU)."""
import os

import numpy as np
import pytest

from bayesianinference_amd import _lib, synthetic as syn
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def close(a, b, n, rtol=1e-8):
    return abs(a - b) <= rtol * max(abs(b), float(n))


@pytest.mark.parametrize("n,d,kernel,world,panel", [(1500, 3, "se_ard", 2, 2), (1500, 3, "matern52_ard", 3, 1),
                                                    (8192, 8, "se_ard", 8, 4), (700, 2, "se", 4, 1),
                                                    (100, 2, "se_ard", 3, 4),      # ONE outer panel: ranks 1, 2 own nothing
                                                    (1, 1, "se", 2, 4),            # N = 1
                                                    (1300, 4, "matern52", 5, 3)])  # world does not divide the panel count
def test_sharded_loglik_matches_oracle_and_single_device(n, d, kernel, world, panel):
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    g = _lib.Handle(X, y, kernel, device=[0] * world)
    info = g.comm_info()
    assert info == {"world": world, "local": world, "comm": "device copies"}
    g.set_option("panel", panel)
    g.set_option("shard_min_n", 0)                       # force the sharded schedule at test sizes
    ll, ld, qd, inf = g.loglik_parts(th)
    assert inf == 0
    h = _lib.Handle(X, y, kernel)
    l1, ld1, qd1, inf1 = h.loglik_parts(th)
    assert inf1 == 0 and close(ll, l1, n, 1e-10) and close(ld, ld1, n, 1e-10) and close(qd, qd1, n, 1e-9)
    if n <= 2000:
        want = orc.log_likelihood(kernel, th, X, y, parts=True)
        assert close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n)
    else:                                                # F3-style scalar from the single-device path (oracle-pinned there)
        assert close(ll, l1, n, 1e-10)
    # repeated calls reuse events / buffers; a second theta; bit-repeatable
    ll2, *_ = g.loglik_parts(th)
    assert ll2 == ll
    g.set_option("bcast_chunks", 0)                      # the panel as ONE broadcast instead of one per tile column: same bits
    ll3, *_ = g.loglik_parts(th)
    g.set_option("bcast_chunks", 1)
    assert ll3 == ll
    th2 = th * 1.07
    assert close(g.loglik(th2)[0], h.loglik(th2)[0], n, 1e-10)
    if n < 8:
        g.close(); h.close()
        return
    # not-SPD verdict travels through the reduction
    bad = th.copy()
    bad[-1] = 0.0
    Xd = X.copy()
    Xd[n // 2] = Xd[3]
    gd = _lib.Handle(Xd, y, kernel, device=[0] * world)
    gd.set_option("shard_min_n", 0)
    gd.set_option("panel", panel)
    assert gd.loglik(bad)[1] == _lib.INFO_NOT_SPD
    assert gd.loglik(np.full_like(th, np.nan))[1] == _lib.INFO_NAN
    assert gd.loglik(th)[1] == 0                          # and the handle keeps working
    gd.close(); g.close(); h.close()


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("n,d,kernel,world,panel", [(1500, 3, "se_ard", 2, 2), (1500, 3, "matern52_ard", 3, 1), (8192, 8, "se_ard", 8, 4),
                                                    (100, 2, "se_ard", 3, 4), (1300, 4, "matern52", 5, 3), (2100, 2, "se_ard", 2, 4)])
def test_sharded_loglik_with_dataflow_panels(n, d, kernel, world, panel, mode):
    """Option dist_panel_df (round 4, the latency-shaped owner path): the owner factors its outer panel as ONE 64-tile dataflow
    launch restricted to the panel's columns instead of three launches per tile column.  Same factorisation up to the summation
    order inside 64-blocks: oracle / single-device values at the usual bars, bit-repeatable, verdicts through the reduction, a
    sharded fit (block inverses rebuilt) and the streamed prediction.  mode 2: the look-ahead update of the panel rides in the
    same launch (the previous panel, read from the receive buffer, is 2 P more slabs of every task).  mode 3 (round 6): mode 2 +
    the launch counts finished tiles per tile column and the owner's communication stream waits for a column's count
    (hipStreamWaitValue32) instead of the launch's end -- the columns of a panel leave while the launch still factors the rest."""
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    g = _lib.Handle(X, y, kernel, device=[0] * world)
    g.set_option("panel", panel)
    g.set_option("shard_min_n", 0)
    g.set_option("dist_panel_df", mode)
    ll, ld, qd, inf = g.loglik_parts(th)
    assert inf == 0 and g.get_option("last_dist_panel_df") == mode
    h = _lib.Handle(X, y, kernel)
    l1, ld1, qd1, _ = h.loglik_parts(th)
    assert close(ll, l1, n, 1e-10) and close(ld, ld1, n, 1e-10) and close(qd, qd1, n, 1e-9)
    if mode == 3:                                        # the library's own choice on this device, and the same arithmetic as 2
        g.set_option("dist_panel_df", -1)
        assert g.loglik_parts(th)[0] == ll and g.get_option("last_dist_panel_df") == 3
        g.set_option("dist_panel_df", 2)
        assert g.loglik_parts(th)[:3] == (ll, ld, qd) and g.get_option("last_dist_panel_df") == 2
        g.set_option("dist_panel_df", 3)
    if n <= 2000:
        want = orc.log_likelihood(kernel, th, X, y, parts=True)
        assert close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n)
    assert g.loglik_parts(th)[0] == ll                   # bit-repeatable
    g.set_option("share_local_panels", 0)                # copies through the receive buffers, like distinct GPUs
    assert close(g.loglik(th * 1.07)[0], h.loglik(th * 1.07)[0], n, 1e-10)
    bad = th.copy()
    bad[0] = np.nan
    assert g.loglik(bad)[1] == _lib.INFO_NAN and g.loglik(th)[1] == 0
    if n >= 1000 and n <= 2000:
        Xs = syn.make_test_points(300, d)
        mo, so = orc.predict_internal(kernel, th, X, y, Xs)
        for replicate in (0, 1):
            g.set_option("replicate_factor", replicate)
            assert g.fit(th) == 0
            mu, var = g.predict(Xs)
            np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(np.sqrt(var), so, rtol=1e-7)
    g.close(); h.close()


@pytest.mark.parametrize("replicate", [0, 1])
def test_sharded_fit_and_predict_shards_test_points(replicate):
    """C3.  replicate_factor = 1: after a sharded fit EVERY rank holds all of L, z and the block
    inverses (each panel is received in place in the rank's dense workspace), test points shard with no further traffic.
    replicate_factor = 0 (default): every rank keeps only ITS panels; prediction streams the factor's panels through the
    ranks once more, each rank substituting its share of the test points; solve / logdet work on the first device."""
    n, d, world = 2300, 4, 4
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("matern52_ard", d)
    Xs = syn.make_test_points(1500, d)                   # >= 2 * 128 * world: sharded over the ranks
    g = _lib.Handle(X, y, "matern52_ard", device=[0] * world)
    g.set_option("shard_min_n", 0)
    g.set_option("panel", 2)
    g.set_option("replicate_factor", replicate)
    g.set_option("share_local_panels", replicate)        # (0: copies through the rotating receive buffers, like distinct GPUs)
    assert g.fit(th) == 0
    mu, var = g.predict(Xs)
    mo, so = orc.predict_internal("matern52_ard", th, X, y, Xs)
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(var), so, rtol=1e-7)
    mu_few, _ = g.predict(Xs[:5])                        # too few to shard: first device alone, same factor
    np.testing.assert_allclose(mu_few, mo[:5], rtol=1e-7, atol=1e-9)
    K = orc.covariance_matrix("matern52_ard", th, X)
    np.testing.assert_allclose(g.solve(y), np.linalg.solve(K, y), rtol=1e-8, atol=1e-9)
    assert close(g.logdet(), np.linalg.slogdet(K)[1], n)
    g.close()


def test_group_handle_deals_batches_and_samples_to_its_devices():
    """Axis (1): thetas / posterior samples are independent units -- contiguous blocks per device, one host
    thread each, no collective.  Below shard_min_n single evaluations stay on the first device."""
    n, d, world = 900, 3, 3
    X, y = syn.make_dataset(n, d)
    g = _lib.Handle(X, y, "se_ard", device=[0] * world)
    h = _lib.Handle(X, y, "se_ard")
    Th = syn.theta_batch(25, "se_ard", d)
    Th[:, -1] = np.maximum(Th[:, -1], 0.05)
    Th[11, 0] = np.nan
    out, info = g.loglik_batch(Th)
    o1, i1 = h.loglik_batch(Th)
    assert np.array_equal(info, i1) and info[11] == _lib.INFO_NAN
    keep = info == 0
    np.testing.assert_allclose(out[keep], o1[keep], rtol=1e-9, atol=1e-9 * n)
    assert close(g.loglik(Th[0])[0], orc.log_likelihood("se_ard", Th[0], X, y), n)     # N < shard_min_n: local
    Xs = syn.make_test_points(60, d)
    mean, var, inf = g.predict_samples(Th[:7], Xs)
    m1, v1, inf1 = h.predict_samples(Th[:7], Xs)
    assert np.array_equal(inf, inf1)
    np.testing.assert_allclose(mean, m1, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(var, v1, rtol=1e-8, atol=1e-12)
    g.close(); h.close()


def test_rank_handle_over_rccl_world_size_one():
    """gphip_create_rank binds RCCL at run time (dlopen) and runs the SAME schedule with ncclBroadcast on the
    comm stream and the 4-double ncclAllReduce; a world of one rank is what a 1-GPU box can host."""
    n, d = 1500, 3
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    cid = _lib.comm_unique_id()
    assert len(cid) == _lib.COMM_ID_BYTES
    g = _lib.Handle(X, y, "se_ard", device=0, rank=0, world=1, comm_id=cid)
    ci = g.comm_info()
    assert ci["world"] == 1 and ci["local"] == 1 and ci["comm"].startswith("rccl (ncclCommInitRank")
    g.set_option("shard_min_n", 0)
    g.set_option("panel", 2)
    ll, ld, qd, inf = g.loglik_parts(th)
    want = orc.log_likelihood("se_ard", th, X, y, parts=True)
    assert inf == 0 and close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n)
    assert g.fit(th) == 0
    Xs = syn.make_test_points(40, d)
    mu, var = g.predict(Xs)
    mo, so = orc.predict_internal("se_ard", th, X, y, Xs)
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
    g.close()


def test_ndev_is_honoured_or_rejected_never_ignored():
    X, y = syn.make_dataset(64, 2)
    with pytest.raises(_lib.GphipError) as e:            # a device that does not exist
        _lib.Handle(X, y, "se_ard", device=[0, 4096])
    assert e.value.status == 5
    with pytest.raises(_lib.GphipError) as e:            # null kernel: K is diagonal, nothing to shard
        _lib.Handle(X, y, "null", device=[0, 0])
    assert e.value.status == 6
    g = _lib.Handle(X, y, "se_ard", device=[0, 0])
    assert g.comm_info()["world"] == 2
    g.close()


def test_gradient_and_cross_covariance_on_a_sharding_group_handle():
    """gphip_loglik_grad on a multi-device handle: the factorisation is sharded (keep = the first device ends up with
    the whole factor and the block inverses), the K^-1 contraction runs on the first device."""
    n, d, world = 1500, 3, 3
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    th[-1] = 0.25
    g = _lib.Handle(X, y, "se_ard", device=[0] * world)
    g.set_option("shard_min_n", 0)
    g.set_option("panel", 2)
    ll, grad, info = g.loglik_grad(th)
    want = orc.log_likelihood_grad("se_ard", th, X, y)
    assert info == 0 and close(ll, orc.log_likelihood("se_ard", th, X, y), n)
    np.testing.assert_allclose(grad, want, rtol=1e-7, atol=1e-7 * np.abs(want).max())
    mu, var = g.predict(X[:4])                           # the factor of theta is still resident after the gradient
    mo, so = orc.predict_internal("se_ard", th, X, y, X[:4])
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
    k, kappa = g.cross_covariance(th, X[:6])
    ko, kap = orc.k_and_kappa("se_ard", th, X, X[:6])
    np.testing.assert_allclose(k, ko, rtol=1e-12, atol=1e-300)
    g.close()


def test_sharding_shards_memory(golden_dir):
    """A rank of a sharding handle keeps only its own block-cyclic panels (+ three receive buffers of one panel each):
    at N = 32768 with 4 ranks and 128-wide panels each rank holds <= 0.3x the single-handle workspace, while the
    evaluation reproduces the ORACLE's scalars for this problem (tests/golden/f3_scalars.npz) and the fit + streamed
    prediction reproduce the single-device handle (the streamed prediction itself is checked against the oracle at
    N = 2300 above); replicate_factor = 1 trades that for the dense workspace on every rank."""
    n, d, world = 32768, 8, 4
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    gold = np.load(os.path.join(golden_dir, "f3_scalars.npz"))
    row = [i for i in range(len(gold["n"])) if int(gold["n"][i]) == n and str(gold["kernel"][i]) == "se_ard"][0]
    Xs = syn.make_test_points(600, d)
    h = _lib.Handle(X, y, "se_ard")
    assert h.fit(th) == 0
    single = h.factor_bytes()
    mu1, var1 = h.predict(Xs)
    h.close()
    g = _lib.Handle(X, y, "se_ard", device=[0] * world)
    g.set_option("shard_min_n", 0)
    g.set_option("panel", 1)
    g.set_option("share_local_panels", 0)                # behave like distinct GPUs: every rank receives into its own buffers
    ll, ld, qd, info = g.loglik_parts(th)
    assert info == 0 and close(ll, float(gold["loglik"][row]), n) and close(ld, float(gold["logdet"][row]), n)
    assert close(qd, float(gold["quad"][row]), n)
    assert g.fit(th) == 0
    mu, var = g.predict(Xs)
    np.testing.assert_allclose(mu, mu1, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(var, var1, rtol=1e-8, atol=1e-12)
    per_rank = [g.factor_bytes(i) for i in range(world)]
    assert max(per_rank) <= 0.3 * single, (per_rank, single)
    assert sum(per_rank) <= 1.2 * single                 # the factor once + the receive buffers
    ll2, info = g.loglik(th)                             # the buffers are reused by the next evaluation
    assert info == 0 and ll2 == ll
    g.set_option("replicate_factor", 1)
    assert g.fit(th) == 0
    assert min(g.factor_bytes(i) for i in range(world)) >= single
    mu2, _ = g.predict(Xs)
    np.testing.assert_allclose(mu2, mu1, rtol=1e-8, atol=1e-9)
    g.close()


def test_distributed_fit_state_does_not_leak_into_later_local_calls():
    """Found by scripts/gpu_api_fuzz.py: after a sharded fit (factor distributed, z gathered into a side vector) a later
    LOCAL call on a member -- batched mixture prediction, a local fit below shard_min_n -- must not read that stale state."""
    n, d, world = 129, 1, 2
    X, y = syn.make_dataset(n, d)
    Th = syn.theta_batch(4, "matern52_ard", d)
    Th[:, -1] = np.maximum(Th[:, -1], 0.2)
    Xs = syn.make_test_points(3, d)
    g = _lib.Handle(X, y, "matern52_ard", device=[0] * world)
    g.set_option("shard_min_n", 0)
    assert g.fit(Th[0]) == 0                              # distributed fit
    g.predict(Xs)
    mS, vS, iS = g.predict_samples(Th, Xs)                # samples dealt to the members: local evaluations
    assert np.all(iS == 0)
    for s in range(4):
        mo, so = orc.predict_internal("matern52_ard", Th[s], X, y, Xs)
        np.testing.assert_allclose(mS[s], mo, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(np.sqrt(vS[s]), so, rtol=1e-7)
    assert g.fit(Th[1]) == 0                              # distributed again ...
    g.set_option("shard_min_n", 1 << 30)
    assert g.fit(Th[2]) == 0                              # ... then a LOCAL fit: prediction must use it, not the stale stream
    mu, var = g.predict(Xs)
    mo, so = orc.predict_internal("matern52_ard", Th[2], X, y, Xs)
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(var), so, rtol=1e-7)
    g.close()


def test_fit_epoch_members_never_serve_an_older_fit():
    """Fit epoch (a group counts its fits, members carry the number of the fit whose factor they hold): a sharded fit
    with all of L on every member (replicate_factor = 1), then a LOCAL fit of the SAME theta with another nugget array
    on the public handle only.  Sharding the test points over the members would now mix two different factors -- the
    theta comparison alone cannot see that; the epoch does, and the prediction runs on the fresh local factor."""
    n, d, world = 700, 2, 2
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    Xs = syn.make_test_points(2 * 128 * world + 40, d)                # enough test points to shard over the members
    rng = np.random.default_rng(5)
    nug_a, nug_b = 0.05 + 0.1 * rng.random(n), 0.4 + 0.3 * rng.random(n)
    g = _lib.Handle(X, y, "se_ard", device=[0] * world)
    g.set_option("replicate_factor", 1)
    g.set_option("shard_min_n", 0)
    assert g.fit_pw(th, None, nug_a) == 0                             # sharded: both members hold L(nug_a)
    mu_a, var_a = g.predict(Xs)
    g.set_option("shard_min_n", 1 << 30)
    assert g.fit_pw(th, None, nug_b) == 0                             # local: member 0 holds L(nug_b), member 1 still L(nug_a)
    mu_b, var_b = g.predict(Xs)
    s = _lib.Handle(X, y, "se_ard")
    assert s.fit_pw(th, None, nug_a) == 0
    ra = s.predict(Xs)
    assert s.fit_pw(th, None, nug_b) == 0
    rb = s.predict(Xs)
    s.close(); g.close()
    np.testing.assert_allclose(mu_a, ra[0], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(var_a, ra[1], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(mu_b, rb[0], rtol=1e-9, atol=1e-11)     # (wrong on the second half of Xs without the epoch)
    np.testing.assert_allclose(var_b, rb[1], rtol=1e-8, atol=1e-11)
    assert np.abs(ra[0] - rb[0]).max() > 1e-4                          # the two fits do differ


@pytest.mark.parametrize("world,panel,mode", [(2, 2, 3), (4, 1, 3), (5, 3, 0), (8, 2, 2)])
def test_owner_yield_only_reorders_independent_updates(world, panel, mode):
    """Option dist_owner_yield (round 6; on from 4 ranks): on the rank that factors panel k + 1 the remainder of REST(k - 1) and
    REST(k) are queued behind the panel launch's end event instead of running beside it.  Every tile still receives its updates
    in the same order, so the results are bit-identical with the option on and off, for every panel form."""
    n, d = 2900, 3
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    g = _lib.Handle(X, y, "se_ard", device=[0] * world)
    g.set_option("panel", panel); g.set_option("shard_min_n", 0); g.set_option("dist_panel_df", mode)
    g.set_option("share_local_panels", 0)
    out = {}
    for yld in (0, 1, -1):
        g.set_option("dist_owner_yield", yld)
        out[yld] = g.loglik_parts(th)
        assert out[yld][3] == 0
    assert out[0] == out[1] == out[-1]
    want = orc.log_likelihood("se_ard", th, X, y, parts=True)
    assert close(out[1][0], want[0], n) and close(out[1][1], want[1], n) and close(out[1][2], want[2], n)
    assert g.fit(th) == 0                                  # a sharded fit keeps the factor where the schedule left it
    Xs = syn.make_test_points(200, d)
    mo, so = orc.predict_internal("se_ard", th, X, y, Xs)
    mu, var = g.predict(Xs)
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(var), so, rtol=1e-7)
    g.close()


@pytest.mark.parametrize("world", [2, 4])
def test_virtual_ranks_side_by_side_on_cu_slices(world, monkeypatch):
    """Developer hook of scripts/gpu_cu_partition.py (read only with GPHIP_TEST_HOOKS=1, which conftest sets): the virtual ranks of
    a one-device group get CU-masked streams on disjoint slices of the chip, so their launches really run at the same time --
    column counters, stream-ordered waits and the owner's yield events under true concurrency.  Same results as the ranks that
    time-slice the whole chip, bit for bit, evaluation after evaluation."""
    n, d = 4200, 3
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    ref = _lib.Handle(X, y, "se_ard", device=[0] * world)
    ref.set_option("shard_min_n", 0); ref.set_option("panel", 2); ref.set_option("share_local_panels", 0)
    want = ref.loglik_parts(th)
    ref.close()
    monkeypatch.setenv("GPHIP_CU_PARTITION", "1")
    g = _lib.Handle(X, y, "se_ard", device=[0] * world)
    monkeypatch.delenv("GPHIP_CU_PARTITION")
    g.set_option("shard_min_n", 0); g.set_option("panel", 2); g.set_option("share_local_panels", 0)
    for yld in (1, 0, -1):
        g.set_option("dist_owner_yield", yld)
        for _ in range(3):
            assert g.loglik_parts(th) == want
    assert want[3] == 0 and close(want[0], orc.log_likelihood("se_ard", th, X, y), n)
    g.close()
