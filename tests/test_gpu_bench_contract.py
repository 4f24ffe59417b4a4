"""bench.py prints ONE JSON line with the keys the driver reads (run here at a small N so it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


@pytest.mark.parametrize("n", [2048, 16384])          # dataflow path / look-ahead path with trailing SYRK launches
def test_bench_json_contract(n):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", str(n), "--steps", "2", "--warmup", "1",
                          "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in REQUIRED:
        assert k in j, k
    assert j["unit"] == "evals/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["higher_is_better"] is True
    assert j["value"] > 0 and abs(j["value"] * j["ms_per_step"] / 1e3 - 1.0) < 1e-6
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["peak"] == 78.6 and abs(r["frac"] * r["peak"] - r["achieved"]) < 1e-9
    if n >= 16384:
        assert r["launches"] > 0 and 20.0 < r["achieved"] < 78.6      # the trailing SYRK ran and was timed
        kb, al = j["roofline_kbuild"], j["roofline_syrk_alone"]
        assert kb["bound"] == "hbm" and kb["peak"] == 8000.0 and 500.0 < kb["achieved"] < 8000.0
        assert kb["algorithmic_per_launch"] == 8.0 * (n * (n + 1) / 2 + n * 8)     # 8 [N(N+1)/2 + N d]
        assert al["bound"] == "mfma" and 20.0 < al["achieved"] < 78.6 and al["launches"] > 0
        assert 0.0 < j["cholesky_frac_of_fp64_mfma_peak"] < 1.0
    assert "workload" in j["config"] and "model" not in j["config"]
