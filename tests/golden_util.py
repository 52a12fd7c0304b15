"""Helpers shared by the parity tests: load a golden case (tests/golden/case_*.npz)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = [
    "pl_test",
    "plpeak",
    "plpeak_full",
    "bspline_test",
    "bspline_iid",
    "bspline_full",
    "bspline_defaults",
    "plpeak_default_tilt",
    "bspline_chieff",
    "bspline_component_masses",
    "bspline_redshift",
    "bspline_redshift_raw",
    "plpeak_smooth",
    "plpeak_iid_spins",
    "bspline_misc",
    "bspline_independent_masses",
    "chm_powerlaw",
    "chm_bspline",
    "gwtc3_pl_test",
    "gwtc3_bspline_test",
]


class GoldenCase:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, f"case_{name}.npz"))
        self.meta = json.loads(str(z["meta"]))
        self.pe = {k[3:]: z[k] for k in z.files if k.startswith("pe/")}
        self.inj = {k[4:]: z[k] for k in z.files if k.startswith("inj/")}
        self.theta = {k[6:]: z[k] for k in z.files if k.startswith("theta/")}
        self.n_points = self.meta["n_points"]
        self.sites = {}
        for k in z.files:
            if k.startswith("sites/"):
                _, fs, site = k.split("/", 2)
                self.sites.setdefault(fs, {})[site] = z[k]
        self.fdgrad = {}
        for k in z.files:
            if k.startswith("fdgrad/"):
                _, i, name_ = k.split("/", 2)
                self.fdgrad.setdefault(int(i), {})[name_] = z[k]
        self.weights_pe = z["weights/pe"]
        self.weights_inj = z["weights/inj"]
        self.composition = self.meta["composition"]
        self.total_inj = self.meta["total_inj"]
        self.nobs = self.meta["nobs"]
        self.tobs = self.meta["tobs"]
        self.flagsets = self.meta["flagsets"]

    def point(self, i):
        return {k: (v[i] if v.ndim > 1 else float(v[i])) for k, v in self.theta.items()}


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    with np.errstate(all="ignore"):
        d = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    d = np.where(a == b, 0.0, d)  # covers +-inf == +-inf and exact zeros
    return float(np.max(d)) if d.size else 0.0


def fd_gradient_tolerance(name, pname):
    """How closely an analytic gradient must match the finite differences of the UNMODIFIED REFERENCE stored in the golden
    cases (relative to max(1, |gradient|)).  The stored differences are the five-point stencil at two steps,
    Richardson-extrapolated (tests/golden/make_golden.py:fd_gradient): measured against the analytic gradients they agree to
    <= 1e-10 for every parameter of every case (worst 9e-11), so the bar is 5e-10.  The one exception is the taper width
    `delta`: log_l is only piecewise smooth in it, its stencil uses a 1e-6 step without extrapolation and is good to ~2e-9."""
    return 1e-8 if pname == "delta" else 5e-10
