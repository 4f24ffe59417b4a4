"""bench.py's multi-rank control flow (launched exactly as the driver launches it: python -m torch.distributed.run,
one process per rank) on a ONE-GPU box: GPHIP_BENCH_BACKEND=gloo puts every rank on device 0 and torch.distributed on
gloo, and the library's own collectives (the `strong` sub-record: one factorisation sharded over all ranks through
gphip_create_rank) go through the tests-only shared-memory collective library, because real RCCL refuses two ranks on
one device.  Checks the ONE JSON line of rank 0: whole-job value, the strong series, and the watchdog that prints the
record without the series when a collective does not return."""
import json
import os
import socket
import subprocess
import sys

import pytest

from bayesianinference_amd import build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(tmp_name, extra, world=2, timeout=600, want_rc=0, npoints=4096):
    fake = build.build_fake_rccl()
    env = dict(os.environ, GPHIP_BENCH_BACKEND="gloo", GPHIP_RCCL_PATH=fake, FAKE_RCCL_SHM=f"/gphip_bench_{os.getpid()}_{tmp_name}",
               LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2",
           "--warmup", "1", "--npoints", str(npoints)] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert (p.returncode == 0) == (want_rc == 0), (p.stdout + p.stderr)[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                       # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_ranks_weak_headline_and_strong_series():
    rec = _run("a", [])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 2 / (rec["ms_per_step"] * 2 * 1e-3)) < 1e-6 * rec["value"]   # whole job: 2 ranks x 2 steps
    assert rec["config"]["parallelism"] == "theta-sharded x2" and rec["dtype"] == "f64"
    st = rec["strong"]
    assert "error" not in st, st
    assert st["scaling"] == "strong" and st["rccl_ranks"] == 2 and st["all_ok"] and st["ms_per_eval"] > 0
    # the split north_star names is also readable at the top level, next to the weak headline
    assert rec["rccl_ranks"] == 2 and rec["strong_ms_per_eval"] == st["ms_per_eval"] and rec["strong_speedup"] > 0
    # the schedule variants only a multi-GPU node can rank are timed beside the default and must agree with it
    assert set(st["variants"]) == {"per_column_broadcast", "dist_panel_df", "owner_yield_on"}
    assert st["default_options"] == {"dist_panel_df": 3, "bcast_two_hop": 0, "dist_owner_yield": 0}
    assert st["oldest_schedule"]["all_ok"] and st["oldest_schedule"]["agrees_with_default"] and st["oldest_schedule"]["ms_per_eval"] > 0
    for v in st["variants"].values():
        assert v["same_results"] and v["ms_per_eval"] > 0
    assert st["best_variant"] in ("default", "per_column_broadcast", "dist_panel_df", "owner_yield_on")
    assert st["variants"]["dist_panel_df"]["same_results"]          # (2 and 3 share the arithmetic: compared bit for bit)
    assert "cpu_baseline" not in rec                                # rank 0 at N = 1 only


def test_strong_series_watchdog_keeps_the_headline():
    # the series cannot finish in 1 ms: the watchdog prints the ONE record and the job ends with a NON-ZERO exit code (a hung
    # collective is not a success)
    rec = _run("b", ["--strong-timeout", "0.001", "--strict-strong"], want_rc=3)
    assert rec["n_gpus"] == 2 and rec["value"] > 0
    assert "no result after" in rec["strong"]["error"] and rec["strong_speedup"] is None
    # without --strict-strong (how the driver launches it) the same situation ends with code 0: the weak headline stands, the
    # failure of the optional series is in the line
    rec = _run("b2", ["--strong-timeout", "0.001"], want_rc=0)
    assert rec["value"] > 0 and "no result after" in rec["strong"]["error"]


def test_eight_ranks_every_schedule_variant_at_sharding_size():
    """The job the driver launches on an 8-GPU node -- 8 ranks, N large enough that the library shards by itself (16384 =
    shard_min_n) -- with every schedule variant of the strong series: the default (one dataflow launch per panel whose tile
    columns are handed to the broadcast stream by counters, from 4 ranks on every message as grouped send / recv + in-place
    all-gather over all links), the same with plain broadcasts, per-tile-column panels with either broadcast form, and dataflow
    panels without the counters with either form.  Eight processes share GPU 0 and the tests-only collective
    library; what is checked is the control flow and that all variants return the same likelihoods."""
    rec = _run("c", ["--strong-timeout", "600"], world=8, timeout=1500, npoints=16384)
    assert rec["n_gpus"] == 8 and rec["value"] > 0 and rec["config"]["parallelism"] == "theta-sharded x8"
    st = rec["strong"]
    assert "error" not in st and "variants_error" not in st, st
    assert st["rccl_ranks"] == 8 and st["all_ok"]
    assert set(st["variants"]) == {"per_column_broadcast", "two_hop", "dist_panel_df", "two_hop_dist_panel_df", "column_signals", "owner_yield_off"}
    assert st["default_options"] == {"dist_panel_df": 3, "bcast_two_hop": 1, "dist_owner_yield": 1}
    assert "model_error" not in st
    for name, v in st["variants"].items():
        assert v["same_results"] and v["ms_per_eval"] > 0, (name, v)
    assert st["two_hop_identical_results"] and st["best_variant"] in {"default", *st["variants"]}
    assert rec["strong_best_variant"] == st["best_variant"] and rec["strong_best_speedup"] >= rec["strong_speedup"] > 0
