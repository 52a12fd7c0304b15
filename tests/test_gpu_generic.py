"""GPU: products of densities that have no ahead-of-time scan kernel, behind the same C ABI.

The reference's user model multiplies whatever densities the user picks (tests/inference_test.py:256-260,
examples/simple_bspline_example.py:58-71).  Products whose sorted term-kind sequence has an ahead-of-time chain run that chain;
every other product of <= GWI_MAX_TERMS terms gets a chain compiled for it at gwi_create (hipRTC: gwinferno_amd/csrc/gwi_jit.h)
and, where hipRTC is missing or switched off (GWI_JIT=0), runs the generic kernel (run-time term loop) -- `gwi_create` never
refuses a model for lack of a kernel.  Both are held against the C oracle (values 1e-9, analytic gradients 1e-8) on three
products written with the drop-in model API that have no ahead-of-time chain; the generic kernel (GWI_FORCE_GENERIC=1) and a
run-time compiled chain (GWI_FORCE_JIT=1) against the ahead-of-time chains on models that do have one.
"""
import os

import numpy as np
import pytest
from golden_util import rel_err

pytestmark = pytest.mark.gpu

VALUE_RTOL = 1e-9


def _compositions():
    from gwinferno_amd import models as M
    from gwinferno_amd.compositions import Composition
    from gwinferno_amd.interpolation import LogXLogYBSpline, LogYBSpline
    from gwinferno_amd.lazy import where_finite

    class PLPeakBetaMagSplineTilt(Composition):
        """PL+Peak m1 x PL q x independent Beta spin magnitudes (parametric.py:71-81) x IID B-spline tilts
        (separable.py:156-218) x PL z: kinds 2,3,4,4,6,7,7."""

        NT = 10
        PARAMS = {"alpha": (), "beta": (), "mpp": (), "sigpp": (), "lam": (), "alpha_a1": (), "beta_a1": (), "alpha_a2": (), "beta_a2": (), "t_coefs": (NT,), "lamb": ()}

        def __init__(self, pe, inj, **kw):
            super().__init__(pe, inj, **kw)
            self.tilt_model = M.BSplineIIDSpinTilts(self.NT, self.pe["cos_tilt_1"], self.pe["cos_tilt_2"], self.inj["cos_tilt_1"], self.inj["cos_tilt_2"], normalize=True)
            self.z_model = M.PowerlawRedshiftModel(self.pe["redshift"], self.inj["redshift"])

        def placeholder(self):
            q = super().placeholder()
            q.update(mpp=30.0, sigpp=5.0, lam=0.1, alpha_a1=2.0, beta_a1=3.0, alpha_a2=2.0, beta_a2=3.0)
            return q

        def weights(self, p, pe_samples):
            d = self.data(pe_samples)
            mass = M.plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], p["alpha"], p["beta"], self.mmin, self.mmax, p["mpp"], p["sigpp"], p["lam"])
            p_a = M.independent_spin_magnitude_beta_dist(d["a_1"], d["a_2"], p["alpha_a1"], p["beta_a1"], p["alpha_a2"], p["beta_a2"])
            return where_finite(mass * p_a * self.tilt_model(p["t_coefs"], pe_samples=pe_samples) * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

        def hypervolume(self, p):
            return self.z_model.normalization(p["lamb"])

        @staticmethod
        def draw(rng):
            return {"alpha": rng.normal(-2.5, 1.0), "beta": rng.normal(1.0, 1.0), "mpp": rng.uniform(20.0, 50.0), "sigpp": rng.uniform(1.0, 10.0), "lam": rng.uniform(0.0, 0.2),
                    "alpha_a1": rng.uniform(1.0, 3.0), "beta_a1": rng.uniform(1.0, 5.0), "alpha_a2": rng.uniform(1.0, 3.0), "beta_a2": rng.uniform(1.0, 5.0),
                    "t_coefs": rng.normal(size=10), "lamb": rng.normal(2.7, 1.0)}

    class SplineMassMixtureTilt(Composition):
        """BSplinePrimaryBSplineRatio (separable.py:446-530) x independent iso+aligned tilt mixtures (parametric.py:93-94) x
        PL z: kinds 5,5,6,7,7."""

        NM, NQ = 12, 7
        PARAMS = {"m1_coefs": (NM,), "q_coefs": (NQ,), "xi1": (), "xi2": (), "sig_t1": (), "sig_t2": (), "lamb": ()}

        def __init__(self, pe, inj, **kw):
            super().__init__(pe, inj, **kw)
            self.mass_model = M.BSplinePrimaryBSplineRatio(self.NM, self.NQ, self.pe["mass_1"], self.inj["mass_1"], self.pe["mass_ratio"], self.inj["mass_ratio"],
                                                           m1min=self.mmin, m2min=self.mmin, mmax=self.mmax, kwargs_m={"basis": LogXLogYBSpline}, kwargs_q={"basis": LogYBSpline})
            self.z_model = M.PowerlawRedshiftModel(self.pe["redshift"], self.inj["redshift"])

        def placeholder(self):
            q = super().placeholder()
            q.update(xi1=0.5, xi2=0.5, sig_t1=1.0, sig_t2=1.0)
            return q

        def weights(self, p, pe_samples):
            d = self.data(pe_samples)
            p_ct = M.independent_spin_tilt(d["cos_tilt_1"], d["cos_tilt_2"], p["xi1"], p["xi2"], p["sig_t1"], p["sig_t2"])
            return where_finite(self.mass_model(p["m1_coefs"], p["q_coefs"], pe_samples=pe_samples) * p_ct * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

        def hypervolume(self, p):
            return self.z_model.normalization(p["lamb"])

        @staticmethod
        def draw(rng):
            return {"m1_coefs": rng.normal(size=12), "q_coefs": rng.normal(size=7), "xi1": rng.uniform(0.0, 1.0), "xi2": rng.uniform(0.0, 1.0), "sig_t1": rng.uniform(0.3, 4.0),
                    "sig_t2": rng.uniform(0.3, 4.0), "lamb": rng.normal(2.7, 1.0)}

    class PowerlawJointTiltSplineSpinsSplineZ(Composition):
        """powerlaw_primary_ratio_pdf (parametric.py:27-30) x default_spin_tilt (:97-102) x independent B-spline spin magnitudes
        (separable.py:82-153) x PowerlawSplineRedshiftModel (spline_perturbation.py:304-372): kinds 1,3,6,7,7,7,10."""

        NA, NZ = 9, 6
        PARAMS = {"alpha": (), "beta": (), "xi": (), "sig_t": (), "a1_coefs": (NA,), "a2_coefs": (NA,), "z_coefs": (NZ,), "lamb": ()}

        def __init__(self, pe, inj, **kw):
            super().__init__(pe, inj, **kw)
            self.mag_model = M.BSplineIndependentSpinMagnitudes(self.NA, self.NA, self.pe["a_1"], self.pe["a_2"], self.inj["a_1"], self.inj["a_2"], normalize=True)
            self.z_model = M.PowerlawSplineRedshiftModel(self.NZ, self.pe["redshift"], self.inj["redshift"])

        def placeholder(self):
            q = super().placeholder()
            q.update(xi=0.5, sig_t=1.0)
            return q

        def weights(self, p, pe_samples):
            d = self.data(pe_samples)
            mass = M.powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=p["alpha"], beta=p["beta"], mmin=self.mmin, mmax=self.mmax)
            p_ct = M.default_spin_tilt(d["cos_tilt_1"], d["cos_tilt_2"], p["xi"], p["sig_t"])
            p_a = self.mag_model(p["a1_coefs"], p["a2_coefs"], pe_samples=pe_samples)
            return where_finite(mass * p_ct * p_a * self.z_model(d["redshift"], p["lamb"], p["z_coefs"]) / d["prior"])

        def hypervolume(self, p):
            return self.z_model.normalization(p["lamb"], p["z_coefs"])

        @staticmethod
        def draw(rng):
            zc = rng.normal(size=6)
            zc[0] = 0.0
            return {"alpha": rng.normal(-2.5, 1.0), "beta": rng.normal(1.0, 1.0), "xi": rng.uniform(0.0, 1.0), "sig_t": rng.uniform(0.3, 4.0), "a1_coefs": rng.normal(size=9),
                    "a2_coefs": rng.normal(size=9), "z_coefs": zc, "lamb": rng.normal(2.7, 1.0)}

    return [PLPeakBetaMagSplineTilt, SplineMassMixtureTilt, PowerlawJointTiltSplineSpinsSplineZ]


@pytest.mark.parametrize("mode", ["jit", "generic"])
@pytest.mark.parametrize("which", [0, 1, 2])
def test_products_without_a_compiled_chain_match_the_c_oracle(which, mode, monkeypatch, tmp_path):
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    if mode == "generic":
        monkeypatch.setenv("GWI_JIT", "0")
    else:
        monkeypatch.setenv("GWI_JIT_CACHE", str(tmp_path))
    pe, inj, total = make_catalog(11, 1300, 9000, seed=31 + which)
    cls = _compositions()[which]
    comp = cls(pe, inj)
    eng = comp.engine()
    info = eng.jit_info()
    if mode == "generic":
        assert eng.scan_kernel_name().startswith("generic"), eng.scan_kernel_name()
        assert not info["compiled_at_run_time"] and "GWI_JIT=0" in info["note"]
    else:
        # a chain of its own, compiled for exactly this kind sequence; dispatched through the AQL queue like an ahead-of-time one
        assert eng.scan_kernel_name().startswith("jit:"), (eng.scan_kernel_name(), info)
        assert info["compiled_at_run_time"] and info["note"] == ""
        assert eng.dispatch_info() == ("disabled by GWI_AQL=0" if os.environ.get("GWI_AQL") == "0" else "aql: active"), eng.dispatch_info()
    orc = COracle(eng.bound)
    rng = np.random.default_rng(5 + which)
    thetas = np.stack([comp.theta(cls.draw(rng)) for _ in range(4)])
    singles = []
    for th in thetas:
        got = eng.evaluate(th, total, min_neff_cut=False)
        ref = orc.evaluate(th, total, min_neff_cut=False)
        assert rel_err(got.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
        assert rel_err(got.log_bfs, ref["logBFs"]) < VALUE_RTOL
        assert rel_err(got.log_neffs, ref["log_nEffs"]) < 1e-8
        assert rel_err(got.summary.log_det_eff, ref["summary"].log_det_eff) < VALUE_RTOL
        assert rel_err(got.norms, ref["norms"]) < VALUE_RTOL
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8
        singles.append(got)
    # flag sets: cuts on; selection marginalised (squared-weight pass through the same kernel)
    for flags in (dict(min_neff_cut=True), dict(min_neff_cut=False, marginalize_selection=True), dict(min_neff_cut=False, max_variance_cut=True)):
        got = eng.evaluate(thetas[0], total, **flags)
        ref = orc.evaluate(thetas[0], total, **flags)
        assert got.log_likelihood == ref["log_likelihood"] or rel_err(got.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8
    # batched launches and the begin / end pair run the same kernel
    batch = eng.evaluate_batch(thetas, total, min_neff_cut=False)
    begin, end = eng.configure_async(total, min_neff_cut=False)
    for k, one in enumerate(singles):
        assert rel_err(batch[k].log_likelihood, one.log_likelihood) < 1e-12
        assert np.allclose(batch[k].grad, one.grad, rtol=1e-10, atol=1e-11)
        begin(thetas[k])
        ll, g = end()
        assert rel_err(ll, one.log_likelihood) < 1e-12 and np.allclose(g, one.grad, rtol=1e-10, atol=1e-11)
    eng.close()


@pytest.mark.parametrize("comp_name", ["plpeak_full", "bspline_full", "bspline_chieff", "chm_powerlaw", "chm_bspline", "plpeak_smooth", "bspline_misc", "plpeak_default_tilt"])
def test_generic_kernel_equals_the_compiled_chain(comp_name, monkeypatch):
    """Every term kind through the generic kernel: models that DO have a compiled chain, forced onto the generic one
    (GWI_FORCE_GENERIC=1), give the compiled chain's value, sites, per-sample log-weights and gradient."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(7, 900, 5000, seed=17)
    rng = np.random.default_rng(3)
    fast = COMPOSITIONS[comp_name](pe, inj)
    ef = fast.engine()
    assert not ef.scan_kernel_name().startswith("generic")
    monkeypatch.setenv("GWI_FORCE_GENERIC", "1")
    slow = COMPOSITIONS[comp_name](pe, inj)
    es = slow.engine()
    assert es.scan_kernel_name().startswith("generic")
    for _ in range(3):
        p = draw_params(comp_name, rng)
        if comp_name == "chm_powerlaw":
            p["mmin"], p["mmax"] = 4.0, 110.0
        th = fast.theta(p)
        a, b = ef.evaluate(th, total, min_neff_cut=False), es.evaluate(th, total, min_neff_cut=False)
        assert rel_err(b.log_likelihood, a.log_likelihood) < 1e-12
        assert rel_err(b.log_bfs, a.log_bfs) < 1e-12 and rel_err(b.log_neffs, a.log_neffs) < 1e-10
        scale = max(1.0, float(np.max(np.abs(a.grad))))
        assert float(np.max(np.abs(b.grad - a.grad))) / scale < 1e-11
        wa, wb = ef.log_weights(th), es.log_weights(th)
        for x, y in zip(wa, wb):
            assert np.array_equal(np.isneginf(x), np.isneginf(y))
            ok = ~np.isneginf(x)
            assert np.max(np.abs(x[ok] - y[ok])) < 1e-11
    ef.close()
    es.close()


@pytest.mark.parametrize("comp_name", ["plpeak", "bspline_iid", "plpeak_full", "bspline_chieff", "chm_bspline"])
def test_run_time_compiled_chain_is_the_ahead_of_time_chain(comp_name, monkeypatch, tmp_path):
    """GWI_FORCE_JIT=1 compiles the chain of a model that has an ahead-of-time one with hipRTC: same template, same flags, same
    headers -- the results agree to 1e-13 (single evaluations through the AQL queue and the HIP stream, batches, log-weights)
    and a second engine finds the code object in the process-wide cache."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(9, 1100, 7000, seed=23)
    rng = np.random.default_rng(11)
    aot = COMPOSITIONS[comp_name](pe, inj).engine()
    assert not aot.scan_kernel_name().startswith(("jit:", "generic"))
    monkeypatch.setenv("GWI_FORCE_JIT", "1")
    monkeypatch.setenv("GWI_JIT_CACHE", str(tmp_path))
    monkeypatch.setenv("GWI_PBATCH_PTS", "3")  # parametric chains: the batch below runs the chain's pbatch kernel (a catalog this small would get one point per row)
    comp = COMPOSITIONS[comp_name](pe, inj)
    jit = comp.engine()
    if "bspline" not in comp_name:
        assert jit.batch_path(5) == "pbatch"
    info = jit.jit_info()
    assert jit.scan_kernel_name().startswith("jit:") and info["compiled_at_run_time"], (jit.scan_kernel_name(), info)
    assert jit.dispatch_info() == ("disabled by GWI_AQL=0" if os.environ.get("GWI_AQL") == "0" else "aql: active")
    assert any(f.endswith(".gwijit") for f in os.listdir(tmp_path)), os.listdir(tmp_path)
    close = dict(rtol=1e-13, atol=1e-13)  # same template, flags and headers; the two compilers' instruction schedules may differ in a contraction
    thetas = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(5)])
    for th in thetas:
        a, b = aot.evaluate(th, total, min_neff_cut=False), jit.evaluate(th, total, min_neff_cut=False)
        assert rel_err(b.log_likelihood, a.log_likelihood) < 1e-13
        assert np.allclose(a.log_bfs, b.log_bfs, **close) and np.allclose(a.log_neffs, b.log_neffs, rtol=1e-11, atol=1e-12)
        assert np.allclose(a.grad, b.grad, rtol=1e-12, atol=1e-12)
        jit.set_timing(2)  # the same through the HIP stream (hipModuleLaunchKernel)
        c = jit.evaluate(th, total, min_neff_cut=False)
        jit.set_timing(0)
        assert c.log_likelihood == b.log_likelihood
        for x, y in zip(aot.log_weights(th), jit.log_weights(th)):
            assert np.array_equal(np.isneginf(x), np.isneginf(y))
            ok = ~np.isneginf(x)
            assert np.allclose(x[ok], y[ok], **close)
    ba, bj = aot.evaluate_batch(thetas, total, min_neff_cut=False), jit.evaluate_batch(thetas, total, min_neff_cut=False)
    for x, y in zip(ba, bj):
        assert rel_err(y.log_likelihood, x.log_likelihood) < 1e-13
        assert np.allclose(x.grad, y.grad, rtol=1e-12, atol=1e-12)
    again = COMPOSITIONS[comp_name](pe, inj).engine()
    assert again.scan_kernel_name() == jit.scan_kernel_name()
    assert again.evaluate(thetas[0], total, min_neff_cut=False).log_likelihood == jit.evaluate(thetas[0], total, min_neff_cut=False).log_likelihood
    for e in (aot, jit, again):
        e.close()


def test_a_damaged_cache_file_is_recompiled(tmp_path):
    """The disk cache is a convenience: a code object the runtime refuses to load (damaged on disk) is deleted and the chain
    compiled again -- the engine still runs its own chain, not the generic kernel."""
    import subprocess
    import sys

    from gwinferno_amd import _native as N

    env = dict(os.environ, GWI_JIT_CACHE=str(tmp_path), GWI_FORCE_JIT="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nimport numpy as np\nfrom gwinferno_amd.compositions import COMPOSITIONS, draw_params\n"
            "from gwinferno_amd.synthetic import make_catalog\npe, inj, total = make_catalog(5, 300, 2000, seed=3)\ncomp = COMPOSITIONS['plpeak'](pe, inj)\n"
            "eng = comp.engine()\nr = eng.evaluate(comp.theta(draw_params('plpeak', np.random.default_rng(1))), total, min_neff_cut=False)\n"
            "print(eng.scan_kernel_name(), eng.jit_info()['from_cache'], repr(r.log_likelihood))" % root)
    first = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert first.returncode == 0, first.stderr[-1500:]
    name, hit, ll = first.stdout.split()[-3:]
    assert name.startswith("jit:") and hit == "False"
    files = [f for f in os.listdir(tmp_path) if f.endswith(".gwijit")]
    assert len(files) == 1
    path = os.path.join(tmp_path, files[0])
    blob = bytearray(open(path, "rb").read())
    start = blob.index(b"\x7fELF")
    blob[start + 64 : start + 4096] = bytes(4096 - 64)  # keep the magic (the cache reader checks that much), wreck the headers behind it
    open(path, "wb").write(bytes(blob))
    second = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert second.returncode == 0, second.stderr[-1500:]
    name2, hit2, ll2 = second.stdout.split()[-3:]
    assert name2 == name and hit2 == "False" and ll2 == ll  # compiled afresh, same answer
    third = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert third.stdout.split()[-3:] == [name, "True", ll]  # ... and the repaired file serves the next process
    assert N is not None


def test_a_model_beyond_the_former_limits(monkeypatch, tmp_path):
    """BASELINE config 5's model (seven B-spline terms x PL z: all EIGHT normalisers the ABI allowed until round 5) times the
    effective-spin model BSplineEffectiveSpinDims (separable.py:706-778: two more normalised linear-spline terms): ten terms,
    nine normalisers, twelve columns, 123 hyper-parameters -- a chain no ahead-of-time kernel exists for (compiled at gwi_create)
    -- against the C oracle."""
    from gwinferno_amd import models as M
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    base = COMPOSITIONS["bspline_full"]

    class FullPlusEffectiveSpins(base):
        NE, NP = 10, 8

        def __init__(self, pe, inj, **kw):
            super().__init__(pe, inj, **kw)
            self.PARAMS = dict(self.PARAMS, e_coefs=(self.NE,), p_coefs=(self.NP,))
            self.chi_model = M.BSplineEffectiveSpinDims(self.NE, self.NP, self.pe["chi_eff"], self.pe["chi_p"], self.inj["chi_eff"], self.inj["chi_p"], normalize=True)

        def placeholder(self):
            q = super().placeholder()
            q["e_coefs"], q["p_coefs"] = np.ones(self.NE), np.ones(self.NP)
            return q

        def spins(self, p, pe_samples):
            return super().spins(p, pe_samples) * self.chi_model(p["e_coefs"], p["p_coefs"], pe_samples=pe_samples)

    monkeypatch.setenv("GWI_JIT_CACHE", str(tmp_path))
    pe, inj, total = make_catalog(9, 900, 6000, seed=61)
    comp = FullPlusEffectiveSpins(pe, inj)
    eng = comp.engine()
    assert len(eng.bound.terms) == 10 and len(eng.bound.norms) == 9 and eng.n_theta == 105 + 18
    assert eng.scan_kernel_name().startswith("jit:6,7,7,7,7,7,7,7,9,9"), eng.scan_kernel_name()
    orc = COracle(eng.bound)
    rng = np.random.default_rng(12)
    for _ in range(3):
        p = draw_params("bspline_full", rng)
        p["e_coefs"], p["p_coefs"] = rng.uniform(0.2, 2.0, size=10), rng.uniform(0.2, 2.0, size=8)
        th = comp.theta(p)
        got, ref = eng.evaluate(th, total, min_neff_cut=False), orc.evaluate(th, total, min_neff_cut=False)
        assert rel_err(got.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
        assert rel_err(got.log_bfs, ref["logBFs"]) < VALUE_RTOL and rel_err(got.norms, ref["norms"]) < VALUE_RTOL
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8
    batch = eng.evaluate_batch(np.stack([th, th]), total, min_neff_cut=False)
    assert rel_err(batch[1].log_likelihood, got.log_likelihood) < 1e-12
    eng.close()


@pytest.mark.parametrize("which", [1, 2])
def test_matrix_core_batched_kernel_compiled_at_run_time(which, monkeypatch, tmp_path):
    """A spline model whose kinds and basis counts have no ahead-of-time instantiation of the batched matrix-core kernel
    (gwi_mfma.h) gets one from hipRTC: forced with GWI_BATCH_MFMA=1 (compiled at gwi_create) it is held against single
    evaluations, the C oracle and itself (bit-reproducible); left alone, the engine compiles it on its first batched launch of
    >= 9 points and uses it from then on (the static rule); with GWI_BATCH_AUTOTUNE=1 it measures it against the 4-tap kernel
    like an ahead-of-time one."""
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    monkeypatch.setenv("GWI_JIT_CACHE", str(tmp_path))
    monkeypatch.setenv("GWI_MAX_BATCH", "24")
    pe, inj, total = make_catalog(11, 1300, 9000, seed=31 + which)
    cls = _compositions()[which]
    rng = np.random.default_rng(50 + which)
    comp = cls(pe, inj)
    thetas = np.stack([comp.theta(cls.draw(rng)) for _ in range(19)])
    monkeypatch.setenv("GWI_BATCH_MFMA", "1")
    forced = cls(pe, inj).engine()
    cal = forced.batch_calibration()
    assert forced.batch_path(16) == "mfma" and cal["matrix_core_kernel"].startswith("compiled jit-mfma:"), cal
    orc = COracle(forced.bound)
    singles = [forced.evaluate(t, total, min_neff_cut=False) for t in thetas]
    for K in (9, 16, 19):
        a, b = forced.evaluate_batch(thetas[:K], total, min_neff_cut=False), forced.evaluate_batch(thetas[:K], total, min_neff_cut=False)
        for k in range(K):
            assert rel_err(a[k].log_likelihood, singles[k].log_likelihood) < 1e-12
            assert np.allclose(a[k].log_bfs, singles[k].log_bfs, rtol=1e-12, atol=1e-12)
            assert np.allclose(a[k].grad, singles[k].grad, rtol=1e-10, atol=1e-11)
            assert a[k].log_likelihood == b[k].log_likelihood and np.array_equal(a[k].grad, b[k].grad)
        ref = orc.evaluate(thetas[0], total, min_neff_cut=False)
        assert rel_err(a[0].log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(a[0].grad - ref["grad"]))) / scale < 1e-8
    forced.close()
    monkeypatch.delenv("GWI_BATCH_MFMA")
    for autotune in (False, True):
        if autotune:
            monkeypatch.setenv("GWI_BATCH_AUTOTUNE", "1")
        auto = cls(pe, inj).engine()
        assert auto.batch_path(16) == "taps" and auto.batch_calibration()["matrix_core_kernel"] == ""  # nothing compiled, nothing measured yet
        batch = auto.evaluate_batch(thetas[:16], total, min_neff_cut=False)
        cal = auto.batch_calibration()
        assert cal["matrix_core_kernel"].startswith("compiled jit-mfma:") and cal["measured"] == autotune
        if autotune:
            assert cal["mfma_us"] > 0 and cal["taps_us"] > 0 and auto.batch_path(16) == ("mfma" if cal["mfma_us"] <= cal["taps_us"] else "taps")
        else:
            assert auto.batch_path(16) == "mfma"
        for k in range(16):
            assert rel_err(batch[k].log_likelihood, singles[k].log_likelihood) < 1e-12 and np.allclose(batch[k].grad, singles[k].grad, rtol=1e-10, atol=1e-11)
        auto.close()
