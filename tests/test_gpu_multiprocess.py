"""The MULTI-PROCESS path of the library -- gphip_create_rank: one process per rank, gphip_loglik / gphip_fit as
collective calls -- with world_size 2 on ONE GPU.  Real RCCL refuses two ranks on the same device, so the two worker
processes (tests/multiproc_worker.py, torch-free) bind a tests-only collective library over POSIX shared memory
(tests/fake_rccl/fake_rccl.cpp) through $GPHIP_RCCL_PATH -- the run-time binding of csrc/rccl_dyn.h.  Everything else
is the product path: per-rank ownership of the outer panels, a rank that owns NOTHING of a step, packing, broadcast
order, unpack-on-receive replication of L, the 4-double all-reduce, verdicts travelling through the reduction, and
each rank predicting its own shard of the test points with no further exchange."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from bayesianinference_amd import build, synthetic as syn
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(a, b, n, rtol=1e-8):
    return abs(a - b) <= rtol * max(abs(b), float(n))


@pytest.mark.parametrize("n,d,kernel,panel,world,two_hop", [
    (1500, 3, "se_ard", 2, 2, 0), (900, 2, "matern52_ard", 1, 3, 0),
    (100, 2, "se_ard", 4, 2, 0),                 # rank 1 owns no panel at all
    (900, 2, "matern52_ard", 1, 3, 1),           # every panel broadcast as scatter (send / recv) + in-place all-gather
    (1300, 3, "se_ard", 2, 4, 1),
    (1500, 3, "se_ard", 2, 3, "dist_panel_df=2"),      # the owner's panel + look-ahead update as ONE dataflow launch reading the
    (2100, 2, "se_ard", 4, 2, "dist_panel_df=2,bcast_two_hop=0"),    # previous panel from the RECEIVE buffer of a real other process
    (2100, 2, "se_ard", 4, 3, "dist_panel_df=3,bcast_two_hop=0"),    # + each column broadcast when the launch's counter says it is final
    (1700, 3, "matern52_ard", 3, 4, "dist_panel_df=3,bcast_two_hop=1")])
def test_rank_handles_in_separate_processes(tmp_path, n, d, kernel, panel, world, two_hop):
    fake = build.build_fake_rccl()
    env = dict(os.environ, GPHIP_NO_TORCH="1", GPHIP_RCCL_PATH=fake, FAKE_RCCL_SHM=f"/gphip_fake_{os.getpid()}_{n}_{two_hop}",
               LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    if isinstance(two_hop, str):
        env["GPHIP_OPTIONS"] = two_hop
        env["FAKE_RCCL_SHM"] = f"/gphip_fake_{os.getpid()}_{n}_{world}_df"
    elif two_hop:
        env["GPHIP_OPTIONS"] = "bcast_two_hop=1"     # (every rank of the job: the setting is part of the agreement check)
    outs = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multiproc_worker.py"), str(r), str(world),
                               outs[r], str(n), str(d), kernel, str(panel)], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung (collective call order differs between ranks?)")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    res = [json.load(open(o)) for o in outs]
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    want = orc.log_likelihood(kernel, th, X, y, parts=True)
    want2 = orc.log_likelihood(kernel, th * 1.05, X, y, parts=True)
    Xs = syn.make_test_points(64, d)
    mo, so = orc.predict_internal(kernel, th, X, y, Xs)
    alpha = np.linalg.solve(orc.covariance_matrix(kernel, th, X), y)
    for r in res:
        assert r["comm"]["world"] == world and r["comm"]["local"] == 1 and r["comm"]["comm"].startswith("rccl (ncclCommInitRank")
        ll, ld, qd, info = r["parts"]
        assert info == 0 and close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n)
        assert r["parts2"][3] == 0 and close(r["parts2"][0], want2[0], n)
        assert r["nan"][1] == 2                                        # the verdict is the same on every rank
        assert r["fit"] == 0 and close(r["logdet"], want[1], n)
        lo, hi = r["shard"]
        np.testing.assert_allclose(r["mu"], mo[lo:hi], rtol=1e-7, atol=1e-9)      # L was replicated on this rank
        np.testing.assert_allclose(np.sqrt(r["var"]), so[lo:hi], rtol=1e-7)
        np.testing.assert_allclose(r["alpha_head"], alpha[:5], rtol=1e-7, atol=1e-9)
    # identical on every rank, bit for bit (same reduction result broadcast back)
    assert all(r["parts"] == res[0]["parts"] for r in res)


@pytest.mark.parametrize("two_hop", [0, 1])
def test_single_process_grouped_rccl_path(tmp_path, two_hop):
    """gphip_create(.., devices, ndev > 1) with RCCL: ncclCommInitAll, one communicator per local rank, the panel
    broadcast as ncclGroupStart / per-rank ncclBroadcast / ncclGroupEnd issued by ONE host thread.  Real RCCL needs
    distinct devices, so on a one-GPU box the run binds the tests-only collective library (in-process mode) and
    GPHIP_COMM=rccl lets virtual ranks use it."""
    n, d, kernel, world, panel = 1500, 3, "se_ard", 3, 2
    fake = build.build_fake_rccl()
    env = dict(os.environ, GPHIP_NO_TORCH="1", GPHIP_RCCL_PATH=fake, GPHIP_COMM="rccl",
               LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    if two_hop:
        env["GPHIP_OPTIONS"] = "bcast_two_hop=1"
    out = str(tmp_path / "inproc.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "inproc_worker.py"), out, str(n), str(d), kernel,
                        str(world), str(panel)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    res = json.load(open(out))
    assert res["comm"]["world"] == world and res["comm"]["local"] == world
    assert res["comm"]["comm"].startswith("rccl (ncclCommInitAll")
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    want = orc.log_likelihood(kernel, th, X, y, parts=True)
    ll, ld, qd, info = res["parts"]
    assert info == 0 and close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n)
    assert res["again"] == res["parts"] and res["fit"] == 0
    mo, so = orc.predict_internal(kernel, th, X, y, syn.make_test_points(700, d))
    np.testing.assert_allclose(res["mu"], mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(res["var"]), so, rtol=1e-7)


@pytest.mark.parametrize("fault,world,fault_rank,arg", [("alloc", 2, 1, 1), ("alloc", 2, 0, 1), ("hip", 3, 1, 7), ("hip", 3, 2, 1),
                                                         ("hip", 2, 0, 25), ("predict", 2, 1, 0), ("predict", 3, 0, 0)])
def test_rank_local_failure_never_leaves_peers_in_a_collective(tmp_path, fault, world, fault_rank, arg):
    """Fault injection on ONE rank of a multi-process job (tests/multiproc_fault_worker.py): a failed slot allocation in
    gphip_dist_begin on a handle that has no layout yet (replicate_factor = 1, share_local_panels = 0), the n-th HIP call of
    the schedule itself (event record / stream wait) failing, a rank whose distributed factor is gone at prediction time.
    Every rank must come back from the call -- with an error -- and the next collective call must work on all of them."""
    n, d, kernel, panel = 900, 3, "se_ard", 2
    fake = build.build_fake_rccl()
    env = dict(os.environ, GPHIP_NO_TORCH="1", GPHIP_RCCL_PATH=fake, FAKE_RCCL_SHM=f"/gphip_fault_{os.getpid()}_{fault}_{fault_rank}",
               LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    outs = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multiproc_fault_worker.py"), str(r), str(world),
                               outs[r], str(n), str(d), kernel, str(panel), fault, str(fault_rank), str(arg)], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung: a local failure left its peers blocked in a collective")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    res = [json.load(open(o)) for o in outs]
    X, y = syn.make_dataset(n, d)
    want = orc.log_likelihood(kernel, syn.default_theta(kernel, d), X, y, parts=True)
    for r in res:
        assert not r["faulted"]["ok"], r                               # EVERY rank reports the failure, from the same call
        assert r["faulted"]["status"] in (3, 4), r
        assert r["after"]["ok"], r                                     # ... and the job carries on
        ll, ld, qd, info = r["after"]["value"]
        assert info == 0 and close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n)
    assert any("another rank" in r["faulted"]["msg"] or "disagree" in r["faulted"]["msg"] for r in res if r["rank"] != fault_rank) or world == 1


@pytest.mark.parametrize("world,bad_rank", [(2, 1), (3, 0)])
def test_create_failure_on_one_rank_fails_every_rank(tmp_path, world, bad_rank):
    """gphip_create_custom_rank with a covariance function that does not compile on ONE rank (tests/multiproc_create_worker.py):
    the failing rank still joins the communicator and the ranks all-reduce a create status, so every rank comes back from the
    create call with an error (the healthy ones: GPHIP_ERR_STATE, 'another rank ..') instead of waiting in ncclCommInitRank for
    ever; the same processes then create a healthy job and evaluate."""
    fake = build.build_fake_rccl()
    env = dict(os.environ, GPHIP_NO_TORCH="1", GPHIP_RCCL_PATH=fake, FAKE_RCCL_SHM=f"/gphip_create_{os.getpid()}_{world}_{bad_rank}",
               LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    outs = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multiproc_create_worker.py"), str(r), str(world), outs[r],
                               str(bad_rank)], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung in the create call of a job whose peer failed")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    res = [json.load(open(o)) for o in outs]
    for r in res:
        assert not r["first"]["ok"], r
        if r["rank"] == bad_rank:
            assert r["first"]["status"] == 1 and "this_symbol_does_not_exist" in r["first"]["msg"], r
        else:
            assert r["first"]["status"] == 4 and "another rank" in r["first"]["msg"], r
    X, y = syn.make_dataset(700, 3)
    want = orc.log_likelihood("se_ard", syn.default_theta("se_ard", 3), X, y, parts=True)
    for r in res:
        ll, ld, qd, info = r["second"]
        assert info == 0 and close(ll, want[0], 700) and close(ld, want[1], 700) and close(qd, want[2], 700)
