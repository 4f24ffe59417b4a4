"""scripts/scale_model.py (CPU): the replay of the sharded schedule on the committed single-GPU step times must keep the
properties DESIGN.md section 8 relies on -- more ranks never slower than fewer for the default, the all-links broadcast never slower
than the plain one from 4 ranks, dataflow panels modelled as final at launch end (not better than the same panels with
per-column readiness), the host-issue term monotone."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("scale_model", os.path.join(ROOT, "scripts", "scale_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_scale_model_properties():
    m = _load()
    data = json.load(open(os.path.join(ROOT, "profiles", "r06_owner_path_32768.json")))
    assert data["N"] == 32768 and {"df0_fuse1", "df2_fuse0"} <= set(data["modes"])
    for alpha, beta in ((10.0, 120.0), (20.0, 60.0)):
        t = {W: m.simulate(data, "df0_fuse1", W, alpha, beta, two_hop=W >= 4)[0] for W in (1, 2, 4, 8)}
        assert t[1] > t[2] > t[4] > t[8] > 0
        plain8 = m.simulate(data, "df0_fuse1", 8, alpha, beta, two_hop=False)[0]
        assert t[8] <= plain8
        end = m.simulate(data, "df2_fuse0", 8, alpha, beta, two_hop=True, final_at_end=True)[0]
        cols = m.simulate(data, "df2_fuse0", 8, alpha, beta, two_hop=True, final_at_end=False)[0]
        assert end >= cols
        # the round-6 default (column counters inside the launch): between "all at the end" and "evenly spread", and a later first
        # column can only cost time
        sig = m.simulate(data, "df2_fuse0", 8, alpha, beta, two_hop=True, first_ready=0.36)[0]
        late = m.simulate(data, "df2_fuse0", 8, alpha, beta, two_hop=True, first_ready=0.9)[0]
        assert cols * 0.999 <= sig <= late <= end * 1.001 and sig <= t[8]
        slow_host = m.simulate(data, "df0_fuse1", 8, alpha, beta, two_hop=True, issue_us=2000.0, one_thread=True)[0]
        assert slow_host >= t[8] and slow_host >= 64 * 2000.0 * 8 * 0.99        # 64 panels x 8 ranks x 2 ms on one thread
    # the one-rank replay is the sum of its steps: no collective, no link
    one = m.simulate(data, "df0_fuse1", 1, 10.0, 120.0)[0]
    steps = data["modes"]["df0_fuse1"]
    assert abs(one - (sum(steps["factor_us"]) + sum(steps["la_us"]) + sum(steps["rest_us"]))) <= 0.15 * one
