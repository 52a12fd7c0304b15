"""Binding of a lazy PE/injection density pair to the HIP engine.

``bind()`` turns the two products built by a user's model function into the flat description of
include/gwi_engine.h (columns, terms, normaliser grids, theta layout); ``NativePopulationLikelihood``
owns the resulting engine handle and exposes value-and-gradient evaluations.  This is host-side
bookkeeping only -- every per-sample operation runs in gwinferno_amd/csrc (HIP, gfx950).
"""
import ctypes as C
import os

import numpy as np

from . import _native as N
from . import expr as E
from .lazy import INJ, PE, Density, LazyNorm, static_key, static_log_expr

NEG_BIG = float(np.nan_to_num(-np.inf))


class BoundModel:
    """Flat model description + data columns, ready for gwi_create."""

    def __init__(self):
        self.col_keys = []      # dedup keys
        self.pe_exprs = []      # one setup expression (gwinferno_amd.expr) per column, kappa last: (N_ev, N_pe)
        self.inj_exprs = []     # ... and for the injection set: (N_inj,)
        self._cols = {}         # host evaluation of the expressions, on first use
        self.terms = []         # dicts mirroring gwi_term
        self.norms = []         # (GridNorm, expo_theta, coef_off)
        self.norm_keys = []
        self.layout = []        # [(factor_index, "scalar", k, offset) | (factor_index, "coefs" | "norm_coefs", n, offset)]
        self.n_theta = 0
        self.kappa_col = -1
        self.vt_norm = -1
        self.n_ev = self.n_pe = self.n_inj = 0
        self.log_const = 0.0    # log of scalar multipliers common to both sides

    def _host_columns(self, side):
        if side not in self._cols:
            exprs, shape = (self.pe_exprs, (self.n_ev, self.n_pe)) if side == PE else (self.inj_exprs, (self.n_inj,))
            self._cols[side] = [np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), shape)) for v in E.evaluate(exprs)]
        return self._cols[side]

    @property
    def pe_cols(self):
        """The PE columns as ``(N_ev, N_pe)`` arrays, evaluated on the HOST (NumPy) on first use: engines without a
        device, the oracles, and the parity test of the device ingest kernel read them."""
        return self._host_columns(PE)

    @property
    def inj_cols(self):
        return self._host_columns(INJ)

    def resident_columns(self, side):
        """What an engine keeps in HBM for this side (``gwi_read_column``): the columns, except that a column read only by
        exponentiated / linear spline terms that agree on their knots holds the KNOT coordinate ``u = (x - lo) / dx`` -- clamped
        into ``[0, n_intervals)`` for bases without the zero-outside flag -- which the engine computes once at creation
        (gwi_engine.hip: spline_knot_kernel; the scan kernels then take interval and fraction from ``(int)u`` and ``fract(u)``)."""
        cols = list(self._host_columns(side))
        knot = (N.TERM_EXP_SPLINE, N.TERM_LINEAR_SPLINE)
        plain = {self.kappa_col}
        for t in self.terms:
            for j, c in enumerate(t["cols"]):
                follows = t["kind"] == N.TERM_POWERLAW_RATIO and (t["flags"] & N.RATIO_LOGM_FROM_SPLINE) and j == 1  # reads the spline's knot coordinate
                if not (t["kind"] in knot and j == 0) and not follows:
                    plain.add(c)
        uses = {}
        for t in self.terms:
            if t["kind"] in knot:
                lo, hi = t["p"][0], t["p"][1]
                n_int = t["n_basis"] - 3
                clamp = t["kind"] == N.TERM_EXP_SPLINE and not (t["flags"] & N.SPLINE_OUTSIDE_ZERO_EXPONENT)
                uses.setdefault(t["cols"][0], set()).add((lo, n_int / (hi - lo), float(np.nextafter(float(n_int), 0.0)) if clamp else -1.0))
        for c, us in uses.items():
            if len(us) == 1 and c not in plain:
                lo, inv_dx, top = next(iter(us))
                u = (cols[c] - lo) * inv_dx
                cols[c] = np.minimum(np.maximum(u, 0.0), top) if top >= 0 else u
        return cols

    def program(self, side, events=None, samples=None):
        """The compiled setup program of one side (``gwi_ingest_program``): ``events`` = (e0, e1) restricts the PE
        sources to a block of events, ``samples`` = (j0, j1) the injection sources to a slice (sharded engines)."""
        exprs = self.pe_exprs if side == PE else self.inj_exprs
        prog = E.compile_program(exprs)
        sl = events if side == PE else samples
        if sl is not None:
            prog.sources = [np.asarray(a)[sl[0]:sl[1]] for a in prog.sources]
        return prog

    def theta_of(self, density):
        """Current hyper-parameter values of ``density`` in this model's theta layout."""
        theta = np.zeros(self.n_theta)
        for fi, what, k, off in self.layout:
            f = density.factors[fi]
            if what == "scalar":
                theta[off] = float(np.asarray(f.scalars[k]))
            elif what == "norm_coefs":
                theta[off : off + k] = np.asarray(f.norm.coefs, dtype=np.float64).ravel()
            else:
                theta[off : off + k] = np.asarray(f.coefs, dtype=np.float64).ravel()
        return theta

    def theta_slices(self, names):
        """Map a list of per-layout-entry names to slices (helper for named compositions)."""
        out = {}
        for name, (fi, what, k, off) in zip(names, self.layout):
            out[name] = slice(off, off + (1 if what == "scalar" else k))
        return out


def structure_key(pe, inj):
    """Hashable description of everything except hyper-parameter VALUES: engines are cached on it."""
    parts = []
    for d in (pe, inj):
        parts.append(tuple(f.structure() + tuple(c.key() for c in f.columns) for f in d.factors))
        parts.append(tuple((sgn, static_key(a)) for sgn, a in d.log_static))
    return tuple(parts)


def bind(pe, inj, hypervolume=None):
    if not isinstance(pe, Density) or not isinstance(inj, Density):
        raise TypeError("weights must be lazy densities produced by gwinferno_amd.models (dense arrays are what the engine replaces)")
    if pe.side not in (PE, None) or inj.side not in (INJ, None):
        raise ValueError("first argument must be the PE-sample product, second the injection product")
    if len(pe.factors) != len(inj.factors):
        raise ValueError("PE and injection products have different numbers of factors")
    if len(pe.factors) > N.GWI_MAX_TERMS:
        raise ValueError(f"at most {N.GWI_MAX_TERMS} factors are supported")
    if abs(pe.log_const - inj.log_const) > 0:
        raise ValueError("PE and injection products carry different constant multipliers")
    bm = BoundModel()
    bm.log_const = pe.log_const

    # ---- shapes from the first column
    first_pe = pe.factors[0].columns[0].expr().shape
    first_inj = inj.factors[0].columns[0].expr().shape
    if len(first_pe) != 2 or len(first_inj) != 1:
        raise ValueError("all PE arrays must share one (N_ev, N_pe) shape and all injection arrays one (N_inj,) shape")
    bm.n_ev, bm.n_pe = first_pe
    bm.n_inj = first_inj[0]
    shapes = {PE: (bm.n_ev, bm.n_pe), INJ: (bm.n_inj,)}

    def check_shape(e, side):
        if e.shape not in ((), shapes[side]):
            raise ValueError("all PE arrays must share one (N_ev, N_pe) shape and all injection arrays one (N_inj,) shape")
        return e

    def add_column(cpe, cinj):
        key = (cpe.key(), cinj.key())
        if key in bm.col_keys:
            return bm.col_keys.index(key)
        epe, einj = cpe.expr(), cinj.expr()
        if epe.shape != shapes[PE] or einj.shape != shapes[INJ]:
            raise ValueError("all PE arrays must share one (N_ev, N_pe) shape and all injection arrays one (N_inj,) shape")
        bm.col_keys.append(key)
        bm.pe_exprs.append(epe)
        bm.inj_exprs.append(einj)
        return len(bm.col_keys) - 1

    # ---- theta layout (shared coefficient vectors -- the IID models -- get one block)
    # canonical term order (sorted by kind, stable): the engine compiles one kernel per kind sequence
    order = sorted(range(len(pe.factors)), key=lambda i: pe.factors[i].kind)
    coef_blocks = {}
    factor_theta = {}
    for fi in order:
        fp, fj = pe.factors[fi], inj.factors[fi]
        if fp.structure() != fj.structure():
            raise ValueError(f"factor {fi}: PE side {fp.structure()} does not match injection side {fj.structure()}")
        slots = []
        for k in range(len(fp.scalars)):
            slots.append(bm.n_theta)
            bm.layout.append((fi, "scalar", k, bm.n_theta))
            bm.n_theta += 1
        coef_off = -1
        if fp.coefs is not None:
            n = int(np.size(fp.coefs))
            if n != fp.n_basis:
                raise ValueError(f"factor {fi}: expected {fp.n_basis} spline coefficients, got {n}")
            key = id(fp.coefs)
            if key in coef_blocks:
                coef_off = coef_blocks[key]
            else:
                coef_off = bm.n_theta
                coef_blocks[key] = coef_off
                bm.layout.append((fi, "coefs", n, coef_off))
                bm.n_theta += n
        factor_theta[fi] = (slots, coef_off)
    if bm.n_theta > N.GWI_MAX_THETA:
        raise ValueError(f"{bm.n_theta} hyper-parameters exceed GWI_MAX_THETA={N.GWI_MAX_THETA}")
    factor_index = {id(f): i for i, f in enumerate(pe.factors)}
    factor_index.update({id(f): i for i, f in enumerate(inj.factors)})

    # ---- kappa = sum of theta-independent log factors, -inf where any static truncation excludes: a setup expression
    # per side (evaluated with the columns, on the device where the engine has one)
    kappa = {}
    for d, side in ((pe, PE), (inj, INJ)):
        kap = E.Sym.const(d.log_const)  # plain scalar multipliers (e.g. the 0.5 of a symmetrised density)
        for sgn, arr in d.log_static:
            e = check_shape(static_log_expr(arr), side)
            kap = kap + e if sgn > 0 else kap - e
        for f in d.factors:
            if f.static_log_expr() is not None:
                kap = kap + check_shape(f.static_log_expr(), side)
        for f in d.factors:
            if f.mask_expr() is not None:
                kap = E.where(check_shape(f.mask_expr(), side), kap, -np.inf)
        # NaN or +inf weights count as zero (tests/inference_test.py:172); a NaN / +inf kappa can
        # only ever produce those
        kappa[side] = E.where(kap < np.inf, kap, -np.inf)

    # ---- a mass-ratio power law next to a spline in log m1 (BSplinePrimaryPowerlawRatio, separable.py:295-365): its `log m1`
    # is an affine map of that spline's knot coordinate, which the engine keeps anyway -- hand the term the spline's column
    # instead of one of its own (include/gwi_engine.h, GWI_RATIO_LOGM_FROM_SPLINE: config 3 then streams 64 B per sample, not 72)
    def logx_source(expr):
        """The array S if ``expr`` is where(valid, log(S), lo) -- the parked coordinate of a log-X spline model -- else None."""
        if expr.op == "where" and expr.args[1].op == "log" and expr.args[1].args[0].op == "src" and expr.args[2].op == "const":
            return expr.args[1].args[0].k
        return None

    logx_splines = {}
    for fi in order:
        fp, fj = pe.factors[fi], inj.factors[fi]
        if fp.kind in (N.TERM_EXP_SPLINE, N.TERM_LINEAR_SPLINE) and fp.mask_expr() is not None and fj.mask_expr() is not None:
            sp, sj = logx_source(fp.columns[0].expr()), logx_source(fj.columns[0].expr())
            if sp is not None and sj is not None:
                logx_splines.setdefault((id(sp), id(sj)), fi)
    folded = {}
    if os.environ.get("GWI_FOLD_LOGM", "1") not in ("", "0"):
        for fi in order:
            fp, fj = pe.factors[fi], inj.factors[fi]
            if fp.kind == N.TERM_POWERLAW_RATIO and len(fp.columns) == 2 and fp.columns[1].transform == "log" and fj.columns[1].transform == "log":
                key = (id(fp.columns[1].source), id(fj.columns[1].source))
                if key in logx_splines:
                    folded[fi] = logx_splines[key]

    # ---- terms
    for fi in order:
        fp, fj = pe.factors[fi], inj.factors[fi]
        extra_flags, extra_p = 0, ()
        if fi in folded:
            sf, sj_ = pe.factors[folded[fi]], inj.factors[folded[fi]]
            lo, hi = sf.consts[0], sf.consts[1]
            n_int = sf.n_basis - 3
            cols = [add_column(fp.columns[0], fj.columns[0]), add_column(sf.columns[0], sj_.columns[0])]
            extra_flags = N.RATIO_LOGM_FROM_SPLINE
            extra_p = (lo, n_int / (hi - lo), (hi - lo) / n_int)
        else:
            cols = [add_column(cp, cj) for cp, cj in zip(fp.columns, fj.columns)]
        slots, coef_off = factor_theta[fi]
        norm_idx = -1
        if fp.norm is not None:
            nkey = (id(fp.norm_owner), fp.tag) if fp.norm_owner is not None else (id(fp), fp.tag)
            if nkey in bm.norm_keys:
                norm_idx = bm.norm_keys.index(nkey)
            else:
                g = fp.norm
                expo_theta = -1
                if g.expo_param is not None:
                    ref_factor, k = g.expo_param
                    expo_theta = factor_theta[factor_index[id(ref_factor)]][0][k]
                norm_coef_off = coef_off
                if g.coefs is not None:  # the normaliser integrates a coefficient vector of its own
                    if int(np.size(g.coefs)) != g.n_basis:
                        raise ValueError(f"factor {fi}: normaliser expects {g.n_basis} coefficients")
                    norm_coef_off = bm.n_theta
                    bm.layout.append((fi, "norm_coefs", g.n_basis, norm_coef_off))
                    bm.n_theta += g.n_basis
                    if bm.n_theta > N.GWI_MAX_THETA:
                        raise ValueError(f"{bm.n_theta} hyper-parameters exceed GWI_MAX_THETA={N.GWI_MAX_THETA}")
                bm.norm_keys.append(nkey)
                bm.norms.append((g, expo_theta, norm_coef_off if g.n_basis > 0 else 0))
                norm_idx = len(bm.norms) - 1
        if fp.kind == N.TERM_PLPEAK_SMOOTH:  # five scalars: gwi_term.theta holds four, coef_off carries the index of delta
            coef_off = slots[4]
        bm.terms.append(dict(kind=fp.kind, cols=cols, theta=slots, n_basis=fp.n_basis, coef_off=max(coef_off, 0), flags=fp.flags | extra_flags, norm=norm_idx,
                             p=tuple(fp.consts) + extra_p,
                             owner=fp.norm_owner))
    if len(bm.norms) > N.GWI_MAX_NORMS:
        raise ValueError(f"{len(bm.norms)} normalisers exceed GWI_MAX_NORMS={N.GWI_MAX_NORMS}")
    # A non-finite column entry (log of a non-positive number, NaN in the data) can only produce a NaN / Inf
    # weight, which counts as zero (tests/inference_test.py:172): exclude the sample and park a finite value
    # in its place, so that the per-sample gradient state the kernel carries stays finite (0 x NaN = NaN).
    for exprs, side in ((bm.pe_exprs, PE), (bm.inj_exprs, INJ)):
        kap = kappa[side]
        for ci, col in enumerate(exprs):
            ok = E.isfinite(col)
            kap = E.where(ok, kap, -np.inf)
            exprs[ci] = E.where(ok, col, 0.0)
        exprs.append(kap)
    bm.kappa_col = len(bm.pe_exprs) - 1
    if len(bm.pe_exprs) > N.GWI_MAX_COLS:
        raise ValueError(f"{len(bm.pe_exprs)} columns exceed GWI_MAX_COLS={N.GWI_MAX_COLS}")

    # ---- which normaliser is the surveyed hypervolume (analysis.py:267)
    if hypervolume is not None:
        if not isinstance(hypervolume, LazyNorm):
            raise TypeError("surveyed_hypervolume must come from z_model.normalization(...)")
        for t in bm.terms:
            if t["norm"] >= 0 and t["owner"] is hypervolume.owner:
                bm.vt_norm = t["norm"]
        if bm.vt_norm < 0:
            raise ValueError("surveyed_hypervolume refers to a model that is not part of the weights")
    return bm


def ingest_columns(outputs, device=-1):
    """Evaluate setup expressions (gwinferno_amd.expr) with the device's ingest kernel (``gwi_ingest_columns``) and
    return them as host arrays: the device twin of :func:`gwinferno_amd.expr.evaluate`."""
    lib = N.load_library()
    prog = E.compile_program(list(outputs))
    shape = next((o.shape for o in outputs if o.shape != ()), ())
    n = int(np.prod(shape)) if shape != () else 1
    st, keep = N.ingest_program(prog)
    cols = [np.empty(n) for _ in outputs]
    ptrs = (N._DP * len(cols))(*[N.as_dp(c) for c in cols])
    rc = lib.gwi_ingest_columns(C.byref(st), n, len(cols), ptrs, int(device))
    del keep
    if rc != 0:
        raise N.NativeEngineError(f"gwi_ingest_columns: {N.STATUS_NAMES.get(rc, rc)}")
    return [c.reshape(shape) for c in cols]


def pin_thread_to_device(device=-1):
    """Restrict the calling thread (and the threads it starts later) to the CPUs next to GPU ``device``
    (``gwi_pin_thread_to_device``; numactl-style host placement, best done before engines are created).  Returns
    whether the affinity was changed."""
    return N.load_library().gwi_pin_thread_to_device(int(device)) == 0


def hbm_bandwidth(device=-1, n_doubles=1 << 27, iters=10):
    """Measured HBM bandwidth of GPU ``device`` in GB/s as ``(read_only, stream_triad)`` (``gwi_hbm_bandwidth``): the figure
    SURVEY section 8(d) asks to report next to the vendor peak the roofline is normalised against.  Arrays of ``n_doubles``
    (default 1 GiB each, far beyond the 256 MB Infinity Cache)."""
    lib = N.load_library()
    r, t = C.c_double(0.0), C.c_double(0.0)
    st = lib.gwi_hbm_bandwidth(int(device), int(n_doubles), int(iters), C.byref(r), C.byref(t))
    if st != 0:
        raise N.NativeEngineError(f"gwi_hbm_bandwidth: {N.STATUS_NAMES.get(st, st)}")
    return r.value, t.value


def shard_bounds(n, rank, world):
    """Contiguous, balanced blocks: the first ``n % world`` ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class EvalResult:
    __slots__ = ("log_likelihood", "grad", "summary", "log_bfs", "log_neffs", "variances", "norms")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


class NativePopulationLikelihood:
    """One catalog + one model structure on one MI355X.

    ``rank`` / ``world`` select this process's shard (contiguous events + injection slice,
    SURVEY.md section 8e); the model objects must have been built from the GLOBAL arrays.
    """

    def __init__(self, pe_density, inj_density, hypervolume=None, device=-1, rank=0, world=1, device_setup=None):
        """``device_setup``: compute the columns (transforms, masks, dVc/dz, kappa) on the GPU from the raw catalog arrays
        (``gwi_create_ingest``; the default wherever the engine has a device) or on the host with NumPy and upload them
        (``gwi_create``; ``GWI_HOST_SETUP=1`` in the environment selects it too -- the two agree to the last bit except
        for the <= 1 ulp of the logarithms)."""
        self.lib = N.load_library()
        self.bound = bm = bind(pe_density, inj_density, hypervolume)
        self.n_theta = bm.n_theta
        self.n_ev_global = bm.n_ev
        self.rank, self.world = rank, world
        e0, e1 = shard_bounds(bm.n_ev, rank, world)
        j0, j1 = shard_bounds(bm.n_inj, rank, world)
        self.event_range, self.inj_range = (e0, e1), (j0, j1)
        self.n_ev, self.n_pe, self.n_inj = e1 - e0, bm.n_pe, j1 - j0
        auto_setup = device_setup is None
        if device_setup is None:
            device_setup = device != N.DEVICE_HOST_ONLY and os.environ.get("GWI_HOST_SETUP", "0") in ("", "0")
        if device_setup and device == N.DEVICE_HOST_ONLY:
            raise ValueError("a host-only handle has no device to set the catalog up on")
        programs = None
        if device_setup:
            try:
                programs = (N.ingest_program(bm.program(PE, events=(e0, e1))), N.ingest_program(bm.program(INJ, samples=(j0, j1))))
            except ValueError as exc:
                # a setup program beyond the ingest kernel's limits (64 registers, 32 sources, 16 tables): the host path
                # computes the same columns with NumPy.  Only an EXPLICIT device_setup=True makes this an error.
                if not auto_setup:
                    raise
                import warnings

                warnings.warn(f"setup program too large for the device ingest kernel ({exc}); computing the catalog columns on the host")
                device_setup = False
        self.device_setup = bool(device_setup)
        pe_cols = inj_cols = ()
        if not device_setup and device != N.DEVICE_HOST_ONLY:
            pe_cols = [N.f64(c[e0:e1]) for c in bm.pe_cols]
            inj_cols = [N.f64(c[j0:j1]) for c in bm.inj_cols]

        spec = N.GwiSpec()
        spec.abi_version = N.GWI_ABI_VERSION
        spec.n_cols = len(bm.pe_exprs)
        spec.kappa_col = bm.kappa_col
        spec.n_theta = bm.n_theta
        spec.n_terms = len(bm.terms)
        spec.n_norms = len(bm.norms)
        spec.vt_norm = bm.vt_norm
        for i, t in enumerate(bm.terms):
            g = spec.terms[i]
            g.kind = t["kind"]
            for k in range(2):
                g.cols[k] = t["cols"][k] if k < len(t["cols"]) else -1
            for k in range(4):
                g.theta[k] = t["theta"][k] if k < len(t["theta"]) else -1
            g.n_basis = t["n_basis"]
            g.coef_off = t["coef_off"]
            g.flags = t["flags"]
            g.norm = t["norm"]
            for k in range(4):
                g.p[k] = t["p"][k] if k < len(t["p"]) else 0.0
        self._keep = [pe_cols, inj_cols]
        for j, (g, expo_theta, coef_off) in enumerate(bm.norms):
            nm = spec.norms[j]
            nm.n_pts = len(g.tw)
            nm.expo_theta = expo_theta
            nm.n_basis = g.n_basis
            nm.coef_off = coef_off
            nm.spline_flags = g.spline_flags
            nm.expo_add = g.expo_add
            nm.lo, nm.hi = g.lo, g.hi
            nm.tw = N.as_dp(g.tw)
            nm.lb = N.as_dp(g.lb)
            nm.l1 = N.as_dp(g.l1)
            nm.us = N.as_dp(g.us)
        handle = C.c_void_p()
        if device_setup:
            (prog_pe, keep_pe), (prog_inj, keep_inj) = programs
            self._keep.append((keep_pe, keep_inj))
            st = self.lib.gwi_create_ingest(C.byref(spec), C.byref(prog_pe), self.n_ev, self.n_pe, C.byref(prog_inj), self.n_inj, device, C.byref(handle))
        else:
            # a host-only handle owns no columns: it gets placeholders it never reads
            dummy = np.zeros(1)
            pe_ptrs = (N._DP * spec.n_cols)(*[N.as_dp(c) for c in (pe_cols or [dummy] * spec.n_cols)])
            inj_ptrs = (N._DP * spec.n_cols)(*[N.as_dp(c) for c in (inj_cols or [dummy] * spec.n_cols)])
            st = self.lib.gwi_create(C.byref(spec), pe_ptrs, self.n_ev, self.n_pe, inj_ptrs, self.n_inj, device, C.byref(handle))
        self.handle = handle
        if st != 0:
            msg = self.lib.gwi_last_error(handle).decode() if handle else "no HIP device visible (gwi_create returned before allocating an engine)"
            if handle:
                self.lib.gwi_destroy(handle)
                self.handle = None
            raise N.NativeEngineError(f"gwi_create failed: {N.STATUS_NAMES.get(st, st)}: {msg}")
        self._keep = None  # the engine copied everything it needs
        self.bytes_per_sample = 8 * spec.n_cols
        self.partial_len = int(self.lib.gwi_partial_len(self.handle))

    # ---------------------------------------------------------------------------------------------
    def _check(self, st):
        if st != 0:
            raise N.NativeEngineError(f"{N.STATUS_NAMES.get(st, st)}: {self.lib.gwi_last_error(self.handle).decode()}")

    def _options(self, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False):
        o = N.GwiOptions()
        o.n_obs = float(self.n_ev_global if nobs is None else nobs)
        o.total_inj = float(total_inj)
        o.marginalize_selection = int(bool(marginalize_selection))
        o.min_neff_cut = int(bool(min_neff_cut))
        o.max_variance_cut = int(bool(max_variance_cut))
        return o

    def _buffers(self):
        """Output buffers + their ctypes pointers, created once (ctypes marshalling is a measurable
        part of a ~30 us evaluation)."""
        b = getattr(self, "_buf", None)
        if b is None:
            b = type("Buf", (), {})()
            b.theta = np.zeros(self.n_theta)
            b.grad = np.zeros(self.n_theta)
            b.lb, b.ln, b.lv = np.zeros(self.n_ev), np.zeros(self.n_ev), np.zeros(self.n_ev)
            b.norms = np.zeros(max(len(self.bound.norms), 1))
            b.summ = N.GwiSummary()
            b.opt = N.GwiOptions()
            b.p_theta, b.p_grad = N.as_dp(b.theta), N.as_dp(b.grad)
            b.p_lb, b.p_ln, b.p_lv, b.p_norms = N.as_dp(b.lb), N.as_dp(b.ln), N.as_dp(b.lv), N.as_dp(b.norms)
            b.r_summ, b.r_opt = C.byref(b.summ), C.byref(b.opt)
            self._buf = b
        return b

    def evaluate(self, theta, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False, want_grad=True, copy=True):
        """Value, gradient and diagnostics of ``hierarchical_likelihood`` at ``theta`` (single device).
        With ``copy=False`` the returned arrays are the engine's reusable buffers (overwritten by the
        next call)."""
        b = self._buffers()
        b.theta[:] = theta
        o = b.opt
        o.n_obs = float(self.n_ev_global if nobs is None else nobs)
        o.total_inj = float(total_inj)
        o.marginalize_selection = int(bool(marginalize_selection))
        o.min_neff_cut = int(bool(min_neff_cut))
        o.max_variance_cut = int(bool(max_variance_cut))
        st = self.lib.gwi_eval(self.handle, b.p_theta, b.r_opt, b.r_summ, b.p_grad if want_grad else None, b.p_lb, b.p_ln, b.p_lv, b.p_norms)
        if st != 0:
            self._check(st)
        n_norms = len(self.bound.norms)
        if copy:
            summ = N.GwiSummary.from_buffer_copy(b.summ)
            return EvalResult(log_likelihood=summ.log_likelihood, grad=b.grad.copy() if want_grad else None, summary=summ, log_bfs=b.lb.copy(), log_neffs=b.ln.copy(),
                              variances=b.lv.copy(), norms=b.norms[:n_norms].copy())
        return EvalResult(log_likelihood=b.summ.log_likelihood, grad=b.grad if want_grad else None, summary=b.summ, log_bfs=b.lb, log_neffs=b.ln, variances=b.lv,
                          norms=b.norms[:n_norms])

    def evaluate_batch(self, thetas, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False, want_grad=True):
        """K hyper-parameter points in one set of launches (vectorised chains).  ``thetas``: (K, n_theta).
        Returns a list of K :class:`EvalResult`."""
        thetas = N.f64(thetas)
        if thetas.ndim != 2 or thetas.shape[1] != self.n_theta:
            raise ValueError(f"thetas must have shape (K, {self.n_theta})")
        K = thetas.shape[0]
        opt = self._options(total_inj, nobs, marginalize_selection, min_neff_cut, max_variance_cut)
        summ = (N.GwiSummary * K)()
        grads = np.zeros((K, self.n_theta)) if want_grad else None
        lb, ln, lv = np.zeros((K, self.n_ev)), np.zeros((K, self.n_ev)), np.zeros((K, self.n_ev))
        n_norms = len(self.bound.norms)
        norms = np.zeros((K, max(n_norms, 1)))
        self._check(self.lib.gwi_eval_batch(self.handle, N.as_dp(thetas), K, C.byref(opt), summ, N.as_dp(grads), N.as_dp(lb), N.as_dp(ln), N.as_dp(lv), N.as_dp(norms)))
        return [EvalResult(log_likelihood=summ[k].log_likelihood, grad=grads[k] if want_grad else None, summary=summ[k], log_bfs=lb[k], log_neffs=ln[k], variances=lv[k],
                           norms=norms[k, :n_norms]) for k in range(K)]

    def pin_thread(self):
        """Restrict the calling thread to the CPUs next to this engine's GPU (``gwi_pin_thread_to_engine``); returns
        whether the affinity was changed (False when sysfs does not describe the device's locality)."""
        return self.lib.gwi_pin_thread_to_engine(self.handle) == 0

    def dispatch_info(self):
        """"aql: active" when plain evaluations go through the engine's own AQL queue, else why they use the HIP stream."""
        return self.lib.gwi_dispatch_info(self.handle).decode()

    def evaluate_sequence(self, thetas, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False, timing_every=0):
        """``len(thetas)`` sequential blocking evaluations in one library call (``gwi_eval_sequence``): the loop a
        sampler runs, without Python between two evaluations.  Returns ``(log_likelihoods, grads[, kernel_ms])``."""
        thetas = N.f64(np.atleast_2d(thetas))
        n = thetas.shape[0]
        if thetas.shape[1] != self.n_theta:
            raise ValueError(f"thetas must be (n, {self.n_theta})")
        opt = self._options(total_inj, nobs, marginalize_selection, min_neff_cut, max_variance_cut)
        ll, grads = np.empty(n), np.empty((n, self.n_theta))
        kms = np.empty((n, 3), dtype=np.float32) if timing_every > 0 else None
        self._check(self.lib.gwi_eval_sequence(self.handle, N.as_dp(thetas), n, C.byref(opt), N.as_dp(ll), N.as_dp(grads), int(timing_every),
                                               kms.ctypes.data_as(C.POINTER(C.c_float)) if kms is not None else None))
        return (ll, grads, kms) if kms is not None else (ll, grads)

    def configure_sequence(self, thetas, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False):
        """:meth:`evaluate_sequence` with everything prepared beforehand: returns ``run() -> (log_likelihoods, grads)`` that is
        ONE library call over the given points (buffers and argument marshalling done here, outside any timed region)."""
        thetas = N.f64(np.atleast_2d(thetas))
        n = thetas.shape[0]
        if thetas.shape[1] != self.n_theta:
            raise ValueError(f"thetas must be (n, {self.n_theta})")
        opt = self._options(total_inj, nobs, marginalize_selection, min_neff_cut, max_variance_cut)
        ll, grads = np.empty(n), np.empty((n, self.n_theta))
        args = (self.handle, N.as_dp(thetas), n, C.byref(opt), N.as_dp(ll), N.as_dp(grads), 0, None)
        fn = self.lib.gwi_eval_sequence

        def run():
            st = fn(*args)
            if st != 0:
                self._check(st)
            return ll, grads

        run._keepalive = (thetas, opt)
        return run

    def configure(self, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False):
        """Fix the likelihood options once; :meth:`value_and_grad` then has the smallest possible
        per-call overhead (what a sampler's inner loop wants)."""
        b = self._buffers()
        o = b.opt
        o.n_obs = float(self.n_ev_global if nobs is None else nobs)
        o.total_inj = float(total_inj)
        o.marginalize_selection = int(bool(marginalize_selection))
        o.min_neff_cut = int(bool(min_neff_cut))
        o.max_variance_cut = int(bool(max_variance_cut))
        fn = self.lib.gwi_eval_sharded if getattr(self, "_comm", False) else self.lib.gwi_eval
        args = (self.handle, b.p_theta, b.r_opt, b.r_summ, b.p_grad, b.p_lb, b.p_ln, b.p_lv, b.p_norms)
        theta_buf, summ, grad = b.theta, b.summ, b.grad

        def value_and_grad(theta):
            theta_buf[:] = theta
            st = fn(*args)
            if st != 0:
                self._check(st)
            return summ.log_likelihood, grad

        self.value_and_grad = value_and_grad
        return value_and_grad

    def configure_async(self, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False):
        """``begin(theta)`` issues the launches of one evaluation and returns at once; ``end()`` waits for it and
        returns ``(log_likelihood, grad)``.  For several engines (one per chain) in flight on one GPU: the
        evaluations of desynchronised chains overlap instead of queueing behind each other's host latency."""
        b = self._buffers()
        o = b.opt
        o.n_obs = float(self.n_ev_global if nobs is None else nobs)
        o.total_inj = float(total_inj)
        o.marginalize_selection = int(bool(marginalize_selection))
        o.min_neff_cut = int(bool(min_neff_cut))
        o.max_variance_cut = int(bool(max_variance_cut))
        lib, handle, theta_buf, summ, grad = self.lib, self.handle, b.theta, b.summ, b.grad
        begin_args = (handle, b.p_theta, b.r_opt, 1)
        end_args = (handle, b.r_summ, b.p_grad, b.p_lb, b.p_ln, b.p_lv, b.p_norms)

        def begin(theta):
            theta_buf[:] = theta
            st = lib.gwi_eval_begin(*begin_args)
            if st != 0:
                self._check(st)

        def end():
            st = lib.gwi_eval_end(*end_args)
            if st != 0:
                self._check(st)
            return summ.log_likelihood, grad

        return begin, end

    def configure_batch(self, k_batch, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False):
        """The batched counterpart of :meth:`configure` for vectorised chains: returns
        ``values_and_grads(thetas[K, n_theta]) -> (log_likelihood[K], grad[K, n_theta])`` writing into
        buffers allocated once (valid until the next call)."""
        K = int(k_batch)
        opt = self._options(total_inj, nobs, marginalize_selection, min_neff_cut, max_variance_cut)
        summ = (N.GwiSummary * K)()
        thetas_buf, grads = np.zeros((K, self.n_theta)), np.zeros((K, self.n_theta))
        values = np.zeros(K)
        args = (self.handle, N.as_dp(thetas_buf), K, C.byref(opt), summ, N.as_dp(grads), None, None, None, None)
        fn = self.lib.gwi_eval_batch
        # log_likelihood is the first double of each gwi_summary: view them without a Python loop
        summ_view = np.frombuffer(summ, dtype=np.float64).reshape(K, -1)[:, 0]

        def values_and_grads(thetas):
            thetas_buf[:] = thetas
            st = fn(*args)
            if st != 0:
                self._check(st)
            values[:] = summ_view
            return values, grads

        self._batch_keepalive = (opt, summ)
        return values_and_grads

    def configure_batch_async(self, k_batch, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False):
        """:meth:`configure_batch` in two halves (``gwi_eval_batch_begin`` / ``gwi_eval_batch_end``): ``begin(thetas[K, n_theta])``
        issues the launches of the K points and returns, ``end() -> (log_likelihood[K], grad[K, n_theta])`` waits for them (buffers
        allocated once, valid until the next ``end``).  Two or three engines driven alternately from ONE thread keep as many sets in
        flight: the scans of the others run while a set is in its combine / final launches and on the host."""
        K = int(k_batch)
        opt = self._options(total_inj, nobs, marginalize_selection, min_neff_cut, max_variance_cut)
        summ = (N.GwiSummary * K)()
        thetas_buf, grads, values = np.zeros((K, self.n_theta)), np.zeros((K, self.n_theta)), np.zeros(K)
        begin_args = (self.handle, N.as_dp(thetas_buf), K, C.byref(opt), 1, 0)
        end_args = (self.handle, summ, N.as_dp(grads), None, None, None, None)
        lib = self.lib
        summ_view = np.frombuffer(summ, dtype=np.float64).reshape(K, -1)[:, 0]

        def begin(thetas):
            thetas_buf[:] = thetas
            st = lib.gwi_eval_batch_begin(*begin_args)
            if st != 0:
                self._check(st)

        def end():
            st = lib.gwi_eval_batch_end(*end_args)
            if st != 0:
                self._check(st)
            values[:] = summ_view
            return values, grads

        self._batch_async_keepalive = (opt, summ, thetas_buf)
        return begin, end

    def configure_callback(self, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False, summary_fields=None):
        """What the NumPyro seam calls once per leapfrog (``likelihood._host_callback``): returns
        ``call(theta) -> (summary[len(summary_fields)], per_event[3, n_ev], grad[n_theta])`` for one point and
        ``call_batch(thetas[K, n_theta]) -> (summary[K, ...], per_event[K, 3, n_ev], grad[K, n_theta])`` for K <= max_batch points
        (one ``gwi_eval_batch``), with the option struct and the argument marshalling done here once.  Every call returns FRESH
        arrays (the caller -- JAX -- may keep them).  ``summary_fields``: names of ``gwi_summary`` members, in the order wanted."""
        fields = [f[0] for f in N.GwiSummary._fields_ if f[0] != "reserved"]
        idx = np.array([fields.index(k) for k in (summary_fields or fields)], dtype=np.intp)
        n_words = C.sizeof(N.GwiSummary) // 8
        opt = self._options(total_inj, nobs, marginalize_selection, min_neff_cut, max_variance_cut)
        r_opt = C.byref(opt)
        lib, handle, n_ev, n_theta = self.lib, self.handle, self.n_ev, self.n_theta
        theta_buf = np.zeros(n_theta)
        p_theta = N.as_dp(theta_buf)
        summ1 = N.GwiSummary()
        view1 = np.frombuffer(summ1, dtype=np.float64)
        r_summ1 = C.byref(summ1)
        dp = N.as_dp

        pe_buf, grad_buf = np.zeros((3, n_ev)), np.zeros(n_theta)
        args1 = (handle, p_theta, r_opt, r_summ1, dp(grad_buf), dp(pe_buf[0]), dp(pe_buf[1]), dp(pe_buf[2]), None)  # (a ctypes pointer costs ~1 us to make: made once)
        eval1 = lib.gwi_eval

        def call(theta):
            theta_buf[:] = theta
            st = eval1(*args1)
            if st != 0:
                self._check(st)
            return view1[idx], pe_buf.copy(), grad_buf.copy()

        batch_state = {}

        def call_batch(thetas):
            K = thetas.shape[0]
            stt = batch_state.get(K)
            if stt is None:
                summ = (N.GwiSummary * K)()
                stt = batch_state[K] = (summ, np.frombuffer(summ, dtype=np.float64).reshape(K, n_words), np.zeros((K, n_theta)))
            summ, view, tbuf = stt
            tbuf[:] = thetas
            lb, ln, lv, grads = np.empty((K, n_ev)), np.empty((K, n_ev)), np.empty((K, n_ev)), np.empty((K, n_theta))
            st = lib.gwi_eval_batch(handle, dp(tbuf), K, r_opt, summ, dp(grads), dp(lb), dp(ln), dp(lv), None)
            if st != 0:
                self._check(st)
            return view[:, idx], np.stack([lb, ln, lv], axis=1), grads

        self._callback_keepalive = (opt, summ1, batch_state, pe_buf, grad_buf, theta_buf)
        return call, call_batch

    def value_and_grad(self, theta):  # replaced by configure()
        raise RuntimeError("call configure(total_inj, ...) first")

    # ---- in-engine RCCL exchange (multi-GPU hot loop without Python/torch in the data path) ----------
    def comm_init(self, unique_id, rank, world, rccl_path=None):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        path = rccl_path.encode() if rccl_path else None
        self._check(self.lib.gwi_comm_init(self.handle, path, buf, int(rank), int(world)))
        self._comm = True

    def shm_comm_init(self, name, rank, world):
        """Attach to the node-local shared-memory segment ``name`` (``gwi_shm_comm_init``): afterwards
        :meth:`evaluate_sharded` / the :meth:`configure` closure exchange the partial records through it -- publish +
        poll between host cores, no collective launch.  Every rank attaches, then one of them unlinks the name."""
        self._check(self.lib.gwi_shm_comm_init(self.handle, name.encode(), int(rank), int(world)))
        self._comm = True

    def shm_exchange(self, record):
        """Publish this rank's record, return all ranks' ``(world, partial_len)`` (host-only handles included)."""
        rec = N.f64(record)
        out = np.zeros((self.world, self.partial_len))
        self._check(self.lib.gwi_shm_exchange(self.handle, N.as_dp(rec), N.as_dp(out)))
        return out

    def evaluate_latencies(self, thetas, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False):
        """Wall-clock seconds of each of ``len(thetas)`` sequential blocking evaluations, measured inside the library
        (``gwi_eval_latencies``)."""
        thetas = N.f64(np.atleast_2d(thetas))
        opt = self._options(total_inj, nobs, marginalize_selection, min_neff_cut, max_variance_cut)
        out = np.zeros(thetas.shape[0])
        self._check(self.lib.gwi_eval_latencies(self.handle, N.as_dp(thetas), thetas.shape[0], C.byref(opt), N.as_dp(out)))
        return out

    def evaluate_sharded(self, theta, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False, want_grad=True, copy=True):
        """Like :meth:`evaluate`, for an engine built with ``rank=/world=`` after :meth:`comm_init`:
        scan of this rank's shard, ncclAllGather of the partial records on the engine's stream,
        identical assembly on every rank.  Per-event arrays are this rank's events."""
        b = self._buffers()
        b.theta[:] = theta
        o = b.opt
        o.n_obs = float(self.n_ev_global if nobs is None else nobs)
        o.total_inj = float(total_inj)
        o.marginalize_selection = int(bool(marginalize_selection))
        o.min_neff_cut = int(bool(min_neff_cut))
        o.max_variance_cut = int(bool(max_variance_cut))
        st = self.lib.gwi_eval_sharded(self.handle, b.p_theta, b.r_opt, b.r_summ, b.p_grad if want_grad else None, b.p_lb, b.p_ln, b.p_lv, b.p_norms)
        if st != 0:
            self._check(st)
        n_norms = len(self.bound.norms)
        if copy:
            summ = N.GwiSummary.from_buffer_copy(b.summ)
            return EvalResult(log_likelihood=summ.log_likelihood, grad=b.grad.copy() if want_grad else None, summary=summ, log_bfs=b.lb.copy(), log_neffs=b.ln.copy(),
                              variances=b.lv.copy(), norms=b.norms[:n_norms].copy())
        return EvalResult(log_likelihood=b.summ.log_likelihood, grad=b.grad if want_grad else None, summary=b.summ, log_bfs=b.lb, log_neffs=b.ln, variances=b.lv,
                          norms=b.norms[:n_norms])

    def eval_partial(self, theta):
        """This rank's partial record (+ local per-event arrays without the global constant)."""
        theta = N.f64(theta)
        rec = np.zeros(self.partial_len)
        lb, ln, lv = np.zeros(self.n_ev), np.zeros(self.n_ev), np.zeros(self.n_ev)
        self._check(self.lib.gwi_eval_partial(self.handle, N.as_dp(theta), N.as_dp(rec), N.as_dp(lb), N.as_dp(ln), N.as_dp(lv)))
        return rec, lb, ln, lv

    def prepare_combine(self, theta):
        """Host-only handles: set the hyper-parameter point whose constants ``combine`` folds in."""
        self._check(self.lib.gwi_prepare_combine(self.handle, N.as_dp(N.f64(theta))))

    def combine(self, records, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False, want_grad=True):
        records = N.f64(records).reshape(-1, self.partial_len)
        opt = self._options(total_inj, nobs, marginalize_selection, min_neff_cut, max_variance_cut)
        summ = N.GwiSummary()
        grad = np.zeros(self.n_theta) if want_grad else None
        norms = np.zeros(max(len(self.bound.norms), 1))
        self._check(self.lib.gwi_combine(self.handle, N.as_dp(records), records.shape[0], C.byref(opt), C.byref(summ), N.as_dp(grad), N.as_dp(norms)))
        return EvalResult(log_likelihood=summ.log_likelihood, grad=grad, summary=summ, log_bfs=None, log_neffs=None, variances=None, norms=norms[: len(self.bound.norms)])

    def launch_geometry(self):
        """``gwi_launch_geometry``: dict of the tile sizes and workgroup counts ``gwi_create`` chose."""
        out = (C.c_int32 * 6)()
        self._check(self.lib.gwi_launch_geometry(self.handle, out))
        return dict(zip(("chunk_pe", "chunk_inj", "tiles_per_event", "n_inj_tiles", "n_scan_blocks", "n_inj_groups"), [int(v) for v in out]))

    def read_column(self, side, col):
        """Column ``col`` of the engine's resident catalog (``gwi_read_column``): ``(n_ev, n_pe)`` for ``side == "pe"``, else
        ``(n_inj,)`` -- what the setup path (host or device) left in HBM."""
        out = np.empty((self.n_ev, self.n_pe) if side == PE else (self.n_inj,))
        self._check(self.lib.gwi_read_column(self.handle, 1 if side == PE else 0, int(col), N.as_dp(out)))
        return out

    def log_weights(self, theta):
        """Per-sample log importance weights (diagnostic; parity with the arrays the reference's
        model function builds, tests/inference_test.py:174-175)."""
        theta = N.f64(theta)
        pe = np.zeros((self.n_ev, self.n_pe))
        inj = np.zeros(self.n_inj)
        self._check(self.lib.gwi_log_weights(self.handle, N.as_dp(theta), N.as_dp(pe), N.as_dp(inj)))
        return pe, inj

    def selftime(self, theta, total_inj, n_iter=1000, min_neff_cut=True):
        """Mean seconds per evaluation of a C-side loop of sequential gwi_eval calls (diagnostic)."""
        opt = self._options(total_inj, None, False, min_neff_cut, False)
        out = np.zeros(1)
        self._check(self.lib.gwi_selftime(self.handle, N.as_dp(N.f64(theta)), C.byref(opt), int(n_iter), N.as_dp(out)))
        return float(out[0])

    def batch_path(self, k_batch=16):
        """"mfma" / "taps" (spline models) or "pbatch" / "rows-per-point" (parametric models): the kernel a batched launch of
        ``k_batch`` points uses (``gwi_batch_path``)."""
        return self.lib.gwi_batch_path(self.handle, int(k_batch)).decode()

    def batch_calibration(self):
        """``{"measured", "mfma_us", "taps_us"}``: the engine's own measurement of its two batched kernels on the first batched
        launch of >= 9 points (``gwi_batch_calibration``); ``measured`` False before that launch or where the path is fixed."""
        m, a, b = C.c_int32(0), C.c_double(0.0), C.c_double(0.0)
        self._check(self.lib.gwi_batch_calibration(self.handle, C.byref(m), C.byref(a), C.byref(b)))
        note = self.lib.gwi_batch_kernel_note(self.handle) if hasattr(self.lib, "gwi_batch_kernel_note") else b""
        return {"measured": bool(m.value), "mfma_us": a.value, "taps_us": b.value, "matrix_core_kernel": (note or b"").decode()}

    def scan_kernel_name(self):
        """The compiled term chain this engine's scan runs, or "generic (run-time term loop)" (``gwi_scan_kernel_name``)."""
        return self.lib.gwi_scan_kernel_name(self.handle).decode()

    def jit_info(self):
        """Whether this engine's scan chain was compiled at ``gwi_create`` (hipRTC, ``gwinferno_amd/csrc/gwi_jit.h``), what that cost
        this process and whether the disk cache supplied the code object; ``note`` says why the generic kernel runs where it
        does (``gwi_jit_info``)."""
        on, sec, hit, note = C.c_int32(0), C.c_double(0.0), C.c_int32(0), C.c_char_p()
        self._check(self.lib.gwi_jit_info(self.handle, C.byref(on), C.byref(sec), C.byref(hit), C.byref(note)))
        return {"compiled_at_run_time": bool(on.value), "compile_seconds": sec.value, "from_cache": bool(hit.value), "note": (note.value or b"").decode()}

    def two_pass_repeats(self):
        """Evaluations this engine had to repeat because a tile's weights lay outside the safe range around its reference
        exponent (``gwi_two_pass_repeats``; at most the first evaluation in ordinary runs)."""
        return int(self.lib.gwi_two_pass_repeats(self.handle))

    def set_timing(self, on=True):
        self._check(self.lib.gwi_set_timing(self.handle, int(on)))

    def last_kernel_ms(self):
        ms = (C.c_float * 3)()
        self._check(self.lib.gwi_last_kernel_ms(self.handle, ms))
        return [float(x) for x in ms]

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gwi_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
