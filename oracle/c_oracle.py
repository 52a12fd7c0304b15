"""ctypes wrapper of oracle/libgwpop_oracle.so (C/OpenMP restatement; TEST INFRASTRUCTURE).  Takes the
flat model description a BoundModel carries (the same gwi_spec the GPU engine receives) and evaluates
value, gradient and sites on the host cores."""
import ctypes as C
import os

import numpy as np

from gwinferno_amd import _native as N

LIB = os.environ.get("GWPOP_ORACLE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgwpop_oracle.so")  # the override: sanitizer builds


def build():
    import subprocess

    subprocess.run(["make", "-C", os.path.dirname(LIB), "-s"], check=True)


class COracle:
    def __init__(self, bound):
        if not os.path.exists(LIB):
            build()
        self.lib = C.CDLL(LIB)
        self.bound = bm = bound
        self.pe_cols = [N.f64(c) for c in bm.pe_cols]
        self.inj_cols = [N.f64(c) for c in bm.inj_cols]
        spec = N.GwiSpec()
        spec.abi_version = N.GWI_ABI_VERSION
        spec.n_cols, spec.kappa_col, spec.n_theta = len(self.pe_cols), bm.kappa_col, bm.n_theta
        spec.n_terms, spec.n_norms, spec.vt_norm = len(bm.terms), len(bm.norms), bm.vt_norm
        for i, t in enumerate(bm.terms):
            g = spec.terms[i]
            g.kind = t["kind"]
            for k in range(2):
                g.cols[k] = t["cols"][k] if k < len(t["cols"]) else 0
            for k in range(4):
                g.theta[k] = t["theta"][k] if k < len(t["theta"]) else 0
            g.n_basis, g.coef_off, g.flags, g.norm = t["n_basis"], t["coef_off"], t["flags"], t["norm"]
            for k in range(4):
                g.p[k] = t["p"][k] if k < len(t["p"]) else 0.0
        self._keep = []
        for j, (g, expo_theta, coef_off) in enumerate(bm.norms):
            nm = spec.norms[j]
            nm.n_pts, nm.expo_theta, nm.n_basis, nm.coef_off, nm.spline_flags = len(g.tw), expo_theta, g.n_basis, coef_off, g.spline_flags
            nm.expo_add, nm.lo, nm.hi = g.expo_add, g.lo, g.hi
            nm.tw, nm.lb, nm.l1, nm.us = N.as_dp(g.tw), N.as_dp(g.lb), N.as_dp(g.l1), N.as_dp(g.us)
        self.spec = spec
        self.pe_ptrs = (N._DP * len(self.pe_cols))(*[N.as_dp(c) for c in self.pe_cols])
        self.inj_ptrs = (N._DP * len(self.inj_cols))(*[N.as_dp(c) for c in self.inj_cols])
        self.lib.gwo_eval.restype = C.c_int
        self.lib.gwo_eval.argtypes = [C.POINTER(N.GwiSpec), C.POINTER(N._DP), C.c_int64, C.c_int64, C.POINTER(N._DP), C.c_int64, N._DP, C.POINTER(N.GwiOptions),
                                      C.POINTER(N.GwiSummary), N._DP, N._DP, N._DP, N._DP, N._DP, C.c_int]
        self.max_threads = int(self.lib.gwo_max_threads())

    def evaluate(self, theta, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False, n_threads=0):
        bm = self.bound
        theta = N.f64(theta)
        opt = N.GwiOptions(float(bm.n_ev if nobs is None else nobs), float(total_inj), int(marginalize_selection), int(min_neff_cut), int(max_variance_cut), 0)
        summ = N.GwiSummary()
        grad = np.zeros(bm.n_theta)
        lb, ln, lv = np.zeros(bm.n_ev), np.zeros(bm.n_ev), np.zeros(bm.n_ev)
        norms = np.zeros(max(len(bm.norms), 1))
        rc = self.lib.gwo_eval(C.byref(self.spec), self.pe_ptrs, bm.n_ev, bm.n_pe, self.inj_ptrs, bm.n_inj, N.as_dp(theta), C.byref(opt), C.byref(summ), N.as_dp(grad),
                               N.as_dp(lb), N.as_dp(ln), N.as_dp(lv), N.as_dp(norms), int(n_threads))
        assert rc == 0
        return {"summary": summ, "log_likelihood": summ.log_likelihood, "grad": grad, "logBFs": lb, "log_nEffs": ln, "variance_log_BFs": lv, "norms": norms[: len(bm.norms)]}
