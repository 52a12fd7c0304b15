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
    """How closely an analytic gradient must match the 4th-order finite differences of the UNMODIFIED REFERENCE stored in
    the golden cases (relative to max(1, |gradient|)).  The differences themselves are good to ~1e-10 wherever log_l is
    smooth over the stencil (measured: every spline coefficient, slope and redshift parameter agrees to <= 3e-11), so the
    bar is 1e-9; the PL+Peak mixture parameters (peak position / width / fraction and the taper) are differenced with a
    step of 1e-3 x |value| across a narrow Gaussian, where the stencil's own truncation error reaches ~6e-8: 2e-7 there."""
    peak_family = ("plpeak", "plpeak_full", "plpeak_default_tilt", "plpeak_smooth", "plpeak_iid_spins", "bspline_misc")
    if name in peak_family and pname in ("mpp", "sigpp", "lam", "alpha", "delta", "beta"):
        return 2e-7
    return 1e-9
