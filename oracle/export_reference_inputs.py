"""Writes tests/golden/reference_inputs.json: the INPUTS of the golden fixtures (X, y, test points, thetas, kernel
and mean names) in a form a Wolfram kernel can Import -- the hand-over file of oracle/make_reference_golden.wl
(TEST INFRASTRUCTURE ONLY).

    python oracle/export_reference_inputs.py

The values are taken from the committed tests/golden/*.npz (doubles written with repr: they round-trip exactly).
Cases: F1 (cfg 1: N=512 d=1 SE), a 96-point prefix of it (minutes -> seconds for the reference's symbolic
covariance build, BGP:45-61), F2 SE-ARD / Matern-5/2-ARD N=256 d=8, F2 constant mean N=333 d=3, and the three
F4 sentinel cases."""
from __future__ import annotations

import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def case(name, kernel, mean, X, y, Xs, thetas, npred):
    return {"name": name, "kernel": kernel, "mean": mean, "X": np.asarray(X).tolist(), "y": np.asarray(y).tolist(),
            "Xs": np.asarray(Xs).tolist(), "thetas": np.asarray(thetas).tolist(), "npred": int(npred)}


def main():
    cases = []
    g = np.load(os.path.join(GOLD, "f1_se_n512_d1.npz"))
    cases.append(case("f1_se_n96_d1", "se", "zero", g["X"][:96], g["y"][:96], g["Xs"][:16], g["thetas"][:6], 2))
    cases.append(case("f1_se_n512_d1", "se", "zero", g["X"], g["y"], g["Xs"], g["thetas"], g["pred_mu"].shape[0]))
    for name in ("f2_se_ard_n256_d8", "f2_matern52_ard_n256_d8", "f2_matern52_const_n333_d3"):
        g = np.load(os.path.join(GOLD, name + ".npz"))
        cases.append(case(name, str(g["kernel"]), str(g["mean"]), g["X"], g["y"], g["Xs"], g["thetas"],
                          g["pred_mu"].shape[0]))
    g = np.load(os.path.join(GOLD, "f4_sentinel.npz"))
    for nm in ("dup", "ill", "ok"):
        cases.append(case("f4_" + nm, "se_ard", "zero", g[nm + "_X"], g[nm + "_y"], g[nm + "_X"][:4],
                          g[nm + "_theta"][None, :], 0))
    path = os.path.join(GOLD, "reference_inputs.json")
    with open(path, "w") as f:
        json.dump({"format": 1, "generator": "oracle/export_reference_inputs.py", "cases": cases}, f)
    print(path, os.path.getsize(path), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
