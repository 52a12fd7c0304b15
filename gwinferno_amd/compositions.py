"""The model compositions of the BASELINE configurations, written against this package's
drop-in model API exactly as a user's NumPyro model function would write them against the
reference.  bench.py, smoke() and the parity tests build engines from these.

Each composition exposes
  PARAMS                    ordered {name: shape} of its hyper-parameters
  weights(p, pe_samples)    lazy importance weights  p(theta | Lambda) / prior
  hypervolume(p)            lazy redshift normaliser passed as ``surveyed_hypervolume``
  engine(...)               NativePopulationLikelihood bound to the catalog
  theta(p)                  flat theta vector in the engine's layout
"""
import numpy as np

from . import models as M
from .engine import NativePopulationLikelihood
from .interpolation import LogXLogYBSpline, LogYBSpline
from .lazy import where_finite


class Composition:
    PARAMS = {}

    def __init__(self, pedict, injdict, mmin=5.0, mmax=100.0):
        self.pe = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in pedict.items()}
        self.inj = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in injdict.items()}
        self.mmin, self.mmax = float(mmin), float(mmax)
        self._engine = None
        self._slices = None

    def data(self, pe_samples):
        return self.pe if pe_samples else self.inj

    def placeholder(self):
        """Any parameter point (used only to establish the model structure)."""
        return {k: (np.zeros(s) if s else 1.0) for k, s in self.PARAMS.items()}

    def engine(self, device=-1, rank=0, world=1, device_setup=None):
        if self._engine is None:
            p = self.placeholder()
            self._engine = NativePopulationLikelihood(self.weights(p, True), self.weights(p, False), self.hypervolume(p), device=device, rank=rank, world=world,
                                                      device_setup=device_setup)
        return self._engine

    def _theta_map(self):
        """For every theta slot, which (parameter name, flat index) feeds it.  Found once by pushing
        uniquely coded parameter values through the model function."""
        if self._slices is None:
            eng = self.engine()
            coded, code = {}, 1048576.015625  # codes no fixed model constant (0.02, 1.0, ...) can collide with; exact in binary
            lookup = {}
            for name, shape in self.PARAMS.items():
                size = int(np.prod(shape)) if shape else 1
                vals = np.arange(code, code + size)
                for i in range(size):
                    lookup[code + i] = (name, i)
                coded[name] = vals.reshape(shape) if shape else float(vals[0])
                code += size
            th = eng.bound.theta_of(self.weights(coded, True))
            # a slot fed by a fixed number (e.g. the constant bounds of a Powerlaw distribution) keeps that number
            self._slices = [lookup.get(v, (None, float(v))) for v in th]
        return self._slices

    def theta(self, p):
        """Flat theta for parameter dict ``p`` (layout = order in which the factors consume parameters)."""
        return np.array([i if name is None else np.asarray(p[name], dtype=np.float64).flat[i] for name, i in self._theta_map()])

    def named_gradient(self, grad, p=None):
        """Scatter a flat gradient back onto parameter names (a parameter feeding several theta
        slots receives the sum, as autodiff would give).  ``p`` (the point) only matters to
        compositions whose theta is a non-trivial function of the parameters."""
        out = {name: np.zeros(int(np.prod(shape)) if shape else 1) for name, shape in self.PARAMS.items()}
        for g, (name, i) in zip(grad, self._theta_map()):
            if name is not None:
                out[name][i] += g
        return {name: (v.reshape(self.PARAMS[name]) if self.PARAMS[name] else float(v[0])) for name, v in out.items()}


class PLTest(Composition):
    """tests/inference_test.py:162-197."""

    PARAMS = {"alpha": (), "beta": (), "lamb": ()}

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.z_model = M.PowerlawRedshiftModel(self.pe["redshift"], self.inj["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        p_m1q = M.powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=p["alpha"], beta=p["beta"], mmin=self.mmin, mmax=self.mmax)
        return where_finite(p_m1q * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class PLPeak(PLTest):
    """BASELINE config 2."""

    PARAMS = {"alpha": (), "beta": (), "mpp": (), "sigpp": (), "lam": (), "lamb": ()}

    def mass(self, p, d):
        return M.plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], p["alpha"], p["beta"], self.mmin, self.mmax, p["mpp"], p["sigpp"], p["lam"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        return where_finite(self.mass(p, d) * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def placeholder(self):
        q = super().placeholder()
        q.update(mpp=30.0, sigpp=5.0, lam=0.1)
        return q


class PLPeakSmooth(PLPeak):
    """PL+Peak with the low-mass taper ``delta`` on both masses (parametric.py:39-53 with delta)."""

    PARAMS = {"alpha": (), "beta": (), "mpp": (), "sigpp": (), "lam": (), "delta": (), "lamb": ()}

    def mass(self, p, d):
        return M.plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], p["alpha"], p["beta"], self.mmin, self.mmax, p["mpp"], p["sigpp"], p["lam"], delta=p["delta"])

    def placeholder(self):
        q = super().placeholder()
        q["delta"] = 4.0
        return q


class PLPeakFull(PLPeak):
    """BASELINE config 1 (examples/simple_powerlaw_peak_example.py:82-94)."""

    PARAMS = {k: () for k in ("alpha", "beta", "mpp", "sigpp", "lam", "alpha_a1", "beta_a1", "alpha_a2", "beta_a2", "xi1", "xi2", "sig_t1", "sig_t2", "lamb")}

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        p_a = M.independent_spin_magnitude_beta_dist(d["a_1"], d["a_2"], p["alpha_a1"], p["beta_a1"], p["alpha_a2"], p["beta_a2"])
        p_ct = M.independent_spin_tilt(d["cos_tilt_1"], d["cos_tilt_2"], p["xi1"], p["xi2"], p["sig_t1"], p["sig_t2"])
        return where_finite(self.mass(p, d) * p_a * p_ct * self.z_model(d["redshift"], p["lamb"]) / d["prior"])


class BSplineTest(Composition):
    """tests/inference_test.py:124-143, 244-285."""

    NM, NQ, NZ = 10, 5, 5

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.NM,), "q_coefs": (self.NQ,), "z_coefs": (self.NZ,), "lamb": ()}
        self.mass_model = M.BSplinePrimaryBSplineRatio(self.NM, self.NQ, self.pe["mass_1"], self.inj["mass_1"], self.pe["mass_ratio"], self.inj["mass_ratio"],
                                                       m1min=self.mmin, m2min=self.mmin, mmax=self.mmax, kwargs_m={"basis": LogXLogYBSpline}, kwargs_q={"basis": LogYBSpline})
        self.z_model = M.PowerlawSplineRedshiftModel(self.NZ, self.pe["redshift"], self.inj["redshift"])

    def spins(self, p, pe_samples):
        return None

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        w = self.mass_model(p["m1_coefs"], p["q_coefs"], pe_samples=pe_samples)
        s = self.spins(p, pe_samples)
        if s is not None:
            w = w * s
        return where_finite(w * self.z_model(d["redshift"], p["lamb"], p["z_coefs"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"], p["z_coefs"])


class BSplineFull(BSplineTest):
    """BASELINE config 5 (examples/simple_bspline_example.py:47-71; pipeline/utils.py:104-155)."""

    NM, NQ, NA, NT, NZ = 30, 14, 12, 12, 12

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.NM,), "q_coefs": (self.NQ,), "a1_coefs": (self.NA,), "a2_coefs": (self.NA,), "t1_coefs": (self.NT,), "t2_coefs": (self.NT,),
                       "z_coefs": (self.NZ,), "lamb": ()}
        self.mag_model = M.BSplineIndependentSpinMagnitudes(self.NA, self.NA, self.pe["a_1"], self.pe["a_2"], self.inj["a_1"], self.inj["a_2"], normalize=True)
        self.tilt_model = M.BSplineIndependentSpinTilts(self.NT, self.NT, self.pe["cos_tilt_1"], self.pe["cos_tilt_2"], self.inj["cos_tilt_1"], self.inj["cos_tilt_2"],
                                                        normalize=True)

    def spins(self, p, pe_samples):
        return self.mag_model(p["a1_coefs"], p["a2_coefs"], pe_samples=pe_samples) * self.tilt_model(p["t1_coefs"], p["t2_coefs"], pe_samples=pe_samples)


class BSplineDefaults(BSplineFull):
    """The reference's DEFAULT B-spline run: ``nspline_dict`` of ``load_pe_and_injections_as_dict`` consumers
    (pipeline/utils.py:29-33: m1 50, q 30, a1 = a2 = 16, tilt1 = tilt2 = 16, redshift 20), ``IID=False``
    (examples/simple_bspline_example.py:50): 165 hyper-parameters."""

    NM, NQ, NA, NT, NZ = 50, 30, 16, 16, 20


class BSplineIID(Composition):
    """BASELINE configs 3/4."""

    NM, NA, NT = 30, 16, 16

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.NM,), "beta": (), "a_coefs": (self.NA,), "t_coefs": (self.NT,), "lamb": ()}
        self.mass_model = M.BSplinePrimaryPowerlawRatio(self.NM, self.pe["mass_1"], self.inj["mass_1"], mmin=self.mmin, mmax=self.mmax)
        self.mag_model = M.BSplineIIDSpinMagnitudes(self.NA, self.pe["a_1"], self.pe["a_2"], self.inj["a_1"], self.inj["a_2"], normalize=True)
        self.tilt_model = M.BSplineIIDSpinTilts(self.NT, self.pe["cos_tilt_1"], self.pe["cos_tilt_2"], self.inj["cos_tilt_1"], self.inj["cos_tilt_2"], normalize=True)
        self.z_model = M.PowerlawRedshiftModel(self.pe["redshift"], self.inj["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        p_m1q = self.mass_model(d["mass_1"], d["mass_ratio"], p["beta"], self.mmin, p["m1_coefs"], pe_samples=pe_samples)
        p_a = self.mag_model(p["a_coefs"], pe_samples=pe_samples)
        p_t = self.tilt_model(p["t_coefs"], pe_samples=pe_samples)
        return where_finite(p_m1q * p_a * p_t * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class PLPeakDefaultTilt(PLPeak):
    """PL+Peak x PL q x default_spin_tilt (parametric.py:97-102) x PL z."""

    PARAMS = {"alpha": (), "beta": (), "mpp": (), "sigpp": (), "lam": (), "xi": (), "sig_t": (), "lamb": ()}

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        p_ct = M.default_spin_tilt(d["cos_tilt_1"], d["cos_tilt_2"], p["xi"], p["sig_t"])
        return where_finite(self.mass(p, d) * p_ct * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def placeholder(self):
        q = super().placeholder()
        q.update(xi=0.5, sig_t=1.0)
        return q


class BSplineChiEff(Composition):
    """BSplinePrimaryBSplineRatio x BSplineEffectiveSpinDims (linear BSpline bases, normalised;
    separable.py:706-778) x PL z."""

    NM, NQ, NE, NP = 12, 8, 10, 8

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.NM,), "q_coefs": (self.NQ,), "e_coefs": (self.NE,), "p_coefs": (self.NP,), "lamb": ()}
        self.mass_model = M.BSplinePrimaryBSplineRatio(self.NM, self.NQ, self.pe["mass_1"], self.inj["mass_1"], self.pe["mass_ratio"], self.inj["mass_ratio"],
                                                       m1min=self.mmin, m2min=self.mmin, mmax=self.mmax)
        self.chi_model = M.BSplineEffectiveSpinDims(self.NE, self.NP, self.pe["chi_eff"], self.pe["chi_p"], self.inj["chi_eff"], self.inj["chi_p"], normalize=True)
        self.z_model = M.PowerlawRedshiftModel(self.pe["redshift"], self.inj["redshift"])

    def placeholder(self):
        q = super().placeholder()
        q["e_coefs"] = np.ones(self.NE)
        q["p_coefs"] = np.ones(self.NP)
        return q

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        w = self.mass_model(p["m1_coefs"], p["q_coefs"], pe_samples=pe_samples) * self.chi_model(p["e_coefs"], p["p_coefs"], pe_samples=pe_samples)
        return where_finite(w * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class BSplineComponentMasses(Composition):
    """BSplineIIDComponentMasses (separable.py:533-613) x PL z."""

    NM = 16

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m_coefs": (self.NM,), "beta": (), "lamb": ()}
        self.mass_model = M.BSplineIIDComponentMasses(self.NM, self.pe["mass_1"], self.pe["mass_2"], self.inj["mass_1"], self.inj["mass_2"], mmin=3.0, mmax=self.mmax)
        self.z_model = M.PowerlawRedshiftModel(self.pe["redshift"], self.inj["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        return where_finite(self.mass_model(p["m_coefs"], beta=p["beta"], pe_samples=pe_samples) * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class PLPeakIIDSpins(PLPeak):
    """PL+Peak x iid_spin_magnitude (parametric.py:67-68, amax = 0.9) x iid_spin_tilt (:89-90) x PL z: one set of spin
    hyper-parameters feeds both components (two theta slots each; the named gradient is their sum)."""

    PARAMS = {k: () for k in ("alpha", "beta", "mpp", "sigpp", "lam", "alpha_a", "beta_a", "xi", "sig_t", "lamb")}
    AMAX = 0.9

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        p_a = M.iid_spin_magnitude(d["a_1"], d["a_2"], p["alpha_a"], p["beta_a"], amax=self.AMAX)
        p_ct = M.iid_spin_tilt(d["cos_tilt_1"], d["cos_tilt_2"], p["xi"], p["sig_t"])
        return where_finite(self.mass(p, d) * p_a * p_ct * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def placeholder(self):
        q = super().placeholder()
        q.update(alpha_a=2.0, beta_a=3.0, xi=0.5, sig_t=1.0)
        return q


class BSplineMisc(Composition):
    """PLPeakPrimaryBSplineRatio (separable.py:368-443) x BSplineSymmetricChiEffective (single.py:233-284) x PL z."""

    NQ, NE = 10, 9

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"alpha": (), "mpp": (), "sigpp": (), "lam": (), "q_coefs": (self.NQ,), "e_coefs": (self.NE,), "lamb": ()}
        self.mass_model = M.PLPeakPrimaryBSplineRatio(self.NQ, self.pe["mass_ratio"], self.inj["mass_ratio"])
        self.chi_model = M.BSplineSymmetricChiEffective(self.NE, self.pe["chi_eff"], self.inj["chi_eff"], normalize=True)
        self.z_model = M.PowerlawRedshiftModel(self.pe["redshift"], self.inj["redshift"])

    def placeholder(self):
        q = super().placeholder()
        q.update(mpp=30.0, sigpp=5.0, lam=0.1, e_coefs=np.ones(self.NE))
        return q

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        mass = self.mass_model(d["mass_1"], p["alpha"], self.mmin, self.mmax, p["mpp"], p["sigpp"], p["lam"], p["q_coefs"], pe_samples=pe_samples)
        return where_finite(mass * self.chi_model(p["e_coefs"], pe_samples=pe_samples) * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class BSplineIndependentMasses(Composition):
    """BSplineIndependentComponentMasses (separable.py:616-703) x PL z."""

    N1, N2 = 14, 11

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.N1,), "m2_coefs": (self.N2,), "beta": (), "lamb": ()}
        self.mass_model = M.BSplineIndependentComponentMasses(self.N1, self.N2, self.pe["mass_1"], self.pe["mass_2"], self.inj["mass_1"], self.inj["mass_2"], mmin1=3.0, mmax1=self.mmax,
                                                              mmin2=3.0, mmax2=self.mmax)
        self.z_model = M.PowerlawRedshiftModel(self.pe["redshift"], self.inj["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        return where_finite(self.mass_model(p["m1_coefs"], p["m2_coefs"], beta=p["beta"], pe_samples=pe_samples) * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class BSplineRedshiftCase(Composition):
    """powerlaw_primary_ratio_pdf x BSplineRedshift(8) with the class defaults (single.py:398-492: LogXBSpline,
    ``normalize=True``).  The engine's theta holds the SCALED exponent coefficients c / (c . I) and, for the
    grid normaliser, the raw ones (models.BSplineRedshift), so theta()/named_gradient() apply that map and
    its chain rule instead of the value-coded lookup of the base class."""

    NZ, KW = 8, {}
    PARAMS = {"alpha": (), "beta": (), "z_coefs": (NZ,)}

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        from .cosmology import planck15_lvk

        cosmo = planck15_lvk()
        self._last_p = None
        self.z_model = M.BSplineRedshift(self.NZ, self.pe["redshift"], self.inj["redshift"], cosmo.dVc_dz(self.pe["redshift"]), cosmo.dVc_dz(self.inj["redshift"]), **self.KW)

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        p_m1q = M.powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=p["alpha"], beta=p["beta"], mmin=self.mmin, mmax=self.mmax)
        return where_finite(p_m1q * self.z_model(np.asarray(p["z_coefs"], dtype=np.float64), pe_samples=pe_samples) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["z_coefs"])

    def placeholder(self):
        q = super().placeholder()
        q["z_coefs"] = np.ones(self.NZ)
        return q

    def theta(self, p):
        self._last_p = p
        return self.engine().bound.theta_of(self.weights(p, True))

    def named_gradient(self, grad, p=None):
        """Gradient with respect to (alpha, beta, z_coefs) from the engine's gradient; the chain rule
        through c' = c / (c . I) needs the point: ``p``, or the one last passed to :meth:`theta`."""
        p = self._last_p if p is None else p
        eng = self.engine()
        out = {}
        slots = {}
        for fi, what, k, off in eng.bound.layout:
            slots.setdefault(what, []).append((off, k))
        scal = [off for off, _ in slots["scalar"]]
        # scalar order follows the sorted term kinds: powerlaw (alpha) then powerlaw-ratio (beta)
        out["alpha"], out["beta"] = float(grad[scal[0]]), float(grad[scal[1]])
        off, k = slots["coefs"][0]
        g_eff = np.asarray(grad[off : off + k])
        integ = self.z_model._basis_integrals
        if integ is None:
            out["z_coefs"] = g_eff.copy()
        else:
            c = np.asarray(p["z_coefs"], dtype=np.float64)
            n = 1.0 / (c @ integ)
            out["z_coefs"] = n * g_eff - n * n * integ * (c @ g_eff)
            noff, nk = slots["norm_coefs"][0]
            out["z_coefs"] = out["z_coefs"] + np.asarray(grad[noff : noff + nk])  # zero: normalisers cancel in log_l
        return out


class BSplineRedshiftRawCase(BSplineRedshiftCase):
    KW = {"normalize": False}


class ChmPowerlaw(Composition):
    """construct_hierarchical_model's weights (analysis.py:401-402) for the model of
    examples/config_files/config.yml: Powerlaw mass_1 with sampled bounds, Powerlaw mass_ratio on [0.02, 1],
    PowerlawRedshift -- written in log space with the distribution classes, as that function sums them."""

    ZMAX, QMIN = 1.9, 0.02
    PARAMS = {"alpha": (), "mmin": (), "mmax": (), "beta": (), "lamb": ()}

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        from .cosmology import planck15_lvk

        self.zgrid = np.linspace(1e-9, self.ZMAX, 1000)  # analysis.py:371-372
        self.dV = planck15_lvk().dVc_dz(self.zgrid)

    def populations(self, p):
        from . import numpyro_distributions as D

        return {"mass_1": D.Powerlaw(p["alpha"], p["mmin"], p["mmax"]), "mass_ratio": D.Powerlaw(p["beta"], self.QMIN, 1.0),
                "redshift": D.PowerlawRedshift(p["lamb"], self.ZMAX, self.zgrid, self.dV)}

    def weights(self, p, pe_samples):
        from .lazy import log

        d = self.data(pe_samples)
        pops = self.populations(p)
        return sum(pops[k].log_prob(d[k]) for k in pops) - log(d["prior"])

    def hypervolume(self, p):
        return self.populations(p)["redshift"].norm

    def placeholder(self):
        return {"alpha": -2.0, "mmin": 5.0, "mmax": 80.0, "beta": 1.0, "lamb": 2.0}


class ChmBSpline(ChmPowerlaw):
    """The same with BSplineDistribution populations (numpyro_distributions.py:266-303) for mass_1 (LogXLogYBSpline
    design matrix on a 1000-point grid) and mass_ratio (LogYBSpline), as tests/numpyro_distributions_test.py:91-129."""

    NM, NQ = 16, 10
    PARAMS = {"m_coefs": (NM,), "q_coefs": (NQ,), "lamb": ()}

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.m_grid = np.linspace(self.mmin, self.mmax, 1000)
        self.m_dmat = LogXLogYBSpline(self.NM, xrange=(self.mmin, self.mmax), normalize=True).bases(self.m_grid)
        self.q_grid = np.linspace(0.0, 1.0, 1000)
        self.q_dmat = LogYBSpline(self.NQ, xrange=(0.0, 1.0), normalize=True).bases(self.q_grid)

    def populations(self, p):
        from . import numpyro_distributions as D

        return {"mass_1": D.BSplineDistribution(self.mmin, self.mmax, p["m_coefs"], self.m_grid, self.m_dmat),
                "mass_ratio": D.BSplineDistribution(0.0, 1.0, p["q_coefs"], self.q_grid, self.q_dmat),
                "redshift": D.PowerlawRedshift(p["lamb"], self.ZMAX, self.zgrid, self.dV)}

    def placeholder(self):
        return Composition.placeholder(self)


COMPOSITIONS = {
    "plpeak_iid_spins": PLPeakIIDSpins,
    "bspline_misc": BSplineMisc,
    "bspline_independent_masses": BSplineIndependentMasses,
    "chm_powerlaw": ChmPowerlaw,
    "chm_bspline": ChmBSpline,
    "plpeak_smooth": PLPeakSmooth,
    "bspline_redshift": BSplineRedshiftCase,
    "bspline_redshift_raw": BSplineRedshiftRawCase,
    "pl_test": PLTest,
    "plpeak": PLPeak,
    "plpeak_full": PLPeakFull,
    "bspline_test": BSplineTest,
    "bspline_iid": BSplineIID,
    "bspline_full": BSplineFull,
    "bspline_defaults": BSplineDefaults,
    "plpeak_default_tilt": PLPeakDefaultTilt,
    "bspline_chieff": BSplineChiEff,
    "bspline_component_masses": BSplineComponentMasses,
}


def draw_params(name, rng):
    """Hyper-parameter draws of SURVEY.md section 8(d)."""
    cls = COMPOSITIONS[name]
    if name == "pl_test":
        return {"alpha": rng.normal(-2.5, 1.0), "beta": rng.normal(1.0, 1.0), "lamb": rng.normal(2.7, 1.0)}
    if name in ("plpeak", "plpeak_full"):
        p = {"alpha": rng.normal(-2.5, 1.0), "beta": rng.normal(1.0, 1.0), "mpp": rng.uniform(20.0, 50.0), "sigpp": rng.uniform(1.0, 10.0), "lam": rng.uniform(0.0, 0.2),
             "lamb": rng.normal(2.7, 1.0)}
        if name == "plpeak_full":
            p.update(alpha_a1=rng.uniform(1.0, 3.0), beta_a1=rng.uniform(1.0, 5.0), alpha_a2=rng.uniform(1.0, 3.0), beta_a2=rng.uniform(1.0, 5.0), xi1=rng.uniform(0.0, 1.0),
                     xi2=rng.uniform(0.0, 1.0), sig_t1=rng.uniform(0.3, 4.0), sig_t2=rng.uniform(0.3, 4.0))
        return {k: p[k] for k in cls.PARAMS}
    if name == "plpeak_iid_spins":
        p = draw_params("plpeak", rng)
        p.update(alpha_a=rng.uniform(1.0, 3.0), beta_a=rng.uniform(1.0, 5.0), xi=rng.uniform(0.0, 1.0), sig_t=rng.uniform(0.3, 4.0))
        return {k: p[k] for k in cls.PARAMS}
    if name == "bspline_misc":
        return {"alpha": rng.normal(-2.5, 1.0), "mpp": rng.uniform(20.0, 50.0), "sigpp": rng.uniform(1.0, 10.0), "lam": rng.uniform(0.0, 0.2),
                "q_coefs": rng.normal(size=cls.NQ), "e_coefs": rng.uniform(0.1, 1.0, size=cls.NE), "lamb": rng.normal(2.7, 1.0)}
    if name == "bspline_independent_masses":
        return {"m1_coefs": rng.normal(size=cls.N1), "m2_coefs": rng.normal(size=cls.N2), "beta": rng.normal(1.0, 1.0), "lamb": rng.normal(2.7, 1.0)}
    if name == "chm_powerlaw":
        return {"alpha": rng.normal(-2.5, 1.0), "mmin": rng.uniform(3.0, 9.0), "mmax": rng.uniform(60.0, 100.0), "beta": rng.normal(1.0, 1.0), "lamb": rng.normal(2.7, 1.0)}
    if name == "chm_bspline":
        return {"m_coefs": rng.normal(size=cls.NM), "q_coefs": rng.normal(size=cls.NQ), "lamb": rng.normal(2.7, 1.0)}
    if name == "plpeak_smooth":
        p = draw_params("plpeak", rng)
        p["delta"] = rng.uniform(1.0, 8.0)
        return {k: p[k] for k in cls.PARAMS}
    if name == "plpeak_default_tilt":
        p = draw_params("plpeak", rng)
        p.update(xi=rng.uniform(0.0, 1.0), sig_t=rng.uniform(0.3, 4.0))
        return {k: p[k] for k in cls.PARAMS}
    if name == "bspline_redshift":
        return {"alpha": rng.normal(-2.5, 1.0), "beta": rng.normal(1.0, 1.0), "z_coefs": rng.uniform(0.3, 2.0, size=cls.NZ)}
    if name == "bspline_redshift_raw":
        return {"alpha": rng.normal(-2.5, 1.0), "beta": rng.normal(1.0, 1.0), "z_coefs": rng.normal(size=cls.NZ)}
    if name == "bspline_chieff":
        return {"m1_coefs": rng.normal(size=12), "q_coefs": rng.normal(size=8), "e_coefs": rng.uniform(0.1, 1.0, size=10), "p_coefs": rng.uniform(0.1, 1.0, size=8),
                "lamb": rng.normal(2.7, 1.0)}
    if name == "bspline_component_masses":
        return {"m_coefs": rng.normal(size=16), "beta": rng.normal(1.0, 1.0), "lamb": rng.normal(2.7, 1.0)}
    shapes = {
        "bspline_test": {"m1_coefs": 10, "q_coefs": 5, "z_coefs": 5},
        "bspline_iid": {"m1_coefs": 30, "a_coefs": 16, "t_coefs": 16},
        "bspline_full": {"m1_coefs": 30, "q_coefs": 14, "a1_coefs": 12, "a2_coefs": 12, "t1_coefs": 12, "t2_coefs": 12, "z_coefs": 12},
        "bspline_defaults": {"m1_coefs": 50, "q_coefs": 30, "a1_coefs": 16, "a2_coefs": 16, "t1_coefs": 16, "t2_coefs": 16, "z_coefs": 20},
    }[name]
    p = {k: rng.normal(size=n) for k, n in shapes.items()}
    if "z_coefs" in p:
        p["z_coefs"][0] = 0.0  # pipeline/utils.py:213-214
    if name == "bspline_iid":
        p["beta"] = rng.normal(1.0, 1.0)
    p["lamb"] = rng.normal(2.7, 1.0)
    return p
