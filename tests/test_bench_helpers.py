"""bench.py's host-side helpers that need no GPU: the replay of the committed PMC summary (the fallback when the counters
cannot be collected in the run), the profiler detection that keeps the live collection from nesting, and the argument
names torchrun's own parser must not choke on."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_pmc_summary_is_replayable():
    tr = _bench().pmc_traffic()
    assert tr is not None and tr["source"].endswith("_pmc_summary.csv")
    # HBM-side bytes per trailing-SYRK launch: algorithmic 2.98 GB, counters 9-14 GB depending on the tile order
    assert 3e9 < tr["bytes_per_launch"] < 2e10


def test_profiler_detection(monkeypatch):
    b = _bench()
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert not b.under_profiler()
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert b.under_profiler()
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert b.under_profiler()


def test_bench_argument_names_survive_torchrun_parser():
    """torch.distributed.run parses with abbreviations allowed: an option of the training script that is a PREFIX of one of
    its own options (--n ...) is rejected as ambiguous before the script ever sees it."""
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    ours = set(re.findall(r'add_argument\("(--[a-z0-9-]+)"', src)) | set(re.findall(r'add_argument\("--[a-z0-9-]+", "(--[a-z0-9-]+)"', src))
    torchrun = ["--nnodes", "--nproc-per-node", "--nproc_per_node", "--rdzv-backend", "--rdzv-endpoint", "--rdzv-id", "--rdzv-conf",
                "--standalone", "--max-restarts", "--monitor-interval", "--start-method", "--event-log-handler", "--role",
                "--module", "--no-python", "--run-path", "--log-dir", "--redirects", "--tee", "--local-ranks-filter",
                "--node-rank", "--master-addr", "--master-port", "--local-addr", "--logs-specs", "--numa-binding",
                "--signals-to-handle", "--virtual-local-rank", "--duplicate-stdout-filters", "--duplicate-stderr-filters"]
    used_by_driver = {"--gpus", "--steps", "--warmup"}
    assert used_by_driver <= ours
    for opt in ours - {"--n", "--d"}:           # (--n / --d stay for direct use; under torchrun pass --npoints / --dim)
        assert not any(t.startswith(opt) and t != opt for t in torchrun), opt
    assert "--npoints" in ours and "--dim" in ours
