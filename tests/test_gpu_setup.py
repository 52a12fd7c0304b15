"""GPU: the setup path on the device against the host (NumPy) evaluation of the
same setup expressions.

The reference prepares masks, logarithms, dVc/dz per sample and the division by the prior eagerly on the host
(models/bsplines/single.py:54-57, parametric.py:113-145, cosmology.py:95-120).  ``gwi_create_ingest`` computes the same
quantities in one HIP kernel from the raw catalog columns.  Bar: every operation except the logarithms reproduces the host
evaluation to the bit; columns that contain a logarithm agree within 2 ulp; kappa's excluded samples (-inf) are the
same samples; engines built either way give the same likelihood.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulp_close(a, b, ulps=2):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(all="ignore"):
        close = np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)))
    return bool(np.all(same | close))


def test_ingest_kernel_reproduces_every_operation_of_the_host_evaluation():
    from gwinferno_amd import expr as E
    from gwinferno_amd.engine import ingest_columns

    rng = np.random.default_rng(11)
    n = 40_000
    x = rng.uniform(-1.0, 4.0, n)
    y = rng.lognormal(0.0, 1.0, n)
    x[:8] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 4.0, -1.0, 1e-320]
    y[8:12] = [np.nan, np.inf, 0.0, 5e-324]
    xf = rng.uniform(0.0, 2.0, n).astype(np.float32)  # a float32 source (the reference's GWTC-3 tensors)
    grid = np.sort(rng.uniform(-0.5, 3.5, 37))
    vals = np.cumsum(rng.normal(size=37))
    x[12:12 + 37] = grid  # exact hits on the table points, the first and the last included
    X, Y, F = E.Sym.src(x), E.Sym.src(y), E.Sym.src(xf)
    exact = [
        X + Y, X - Y, X * Y, X / Y, -X, abs(X), E.sqrt(Y), X * Y - 0.3, (X * X) * X + 2.0 * Y,            # arithmetic, never fused
        E.where(~((X < 0.5) | (X > 2.5)), X, -np.inf), E.where((X <= 1.0) & (Y >= 1.0), X, Y),            # comparisons, logic, selection
        E.where(E.isfinite(X / Y), X / Y, 0.0), E.where(X < np.inf, X, -np.inf),                           # the guards bind() applies
        E.interp(X, grid, vals), E.gridindex(X, grid), E.interp(F, grid, vals) * F, F + 1.0,               # tables, float32 source
        4 * np.pi * (E.interp(Y, grid, vals) * E.interp(Y, grid, vals)) * (4415.2 / E.sqrt(0.69 + 0.31 * ((1.0 + Y) * (1.0 + Y) * (1.0 + Y)))),
    ]
    logs = [E.log(Y), E.log(X), E.log1p(X), E.log(1.0 - X / 3.0), -E.log(Y), E.log(X / Y) * 3.0]
    host = E.evaluate(exact + logs)
    dev = ingest_columns(exact[:9] + logs[:3]) + ingest_columns(exact[9:] + logs[3:])  # <= GWI_MAX_COLS outputs per program
    host = host[:9] + host[len(exact):len(exact) + 3] + host[9:len(exact)] + host[len(exact) + 3:]
    k = 0
    for part, (ne, nl) in enumerate(((9, 3), (len(exact) - 9, 3))):
        for j in range(ne + nl):
            h, d = np.asarray(host[k], dtype=np.float64), dev[k]
            if j < ne:
                assert np.array_equal(h, d, equal_nan=True), (part, j, np.flatnonzero(~((h == d) | (np.isnan(h) & np.isnan(d))))[:5])
            else:
                assert np.array_equal(np.isnan(h), np.isnan(d)) and np.array_equal(np.isinf(h), np.isinf(d)), (part, j)
                assert _ulp_close(h, d), (part, j)
            k += 1
    # a difference of logarithms (BSplineRedshift's log dVc/dz - log1p z): each within an ulp, the difference within an ulp of the operands
    a, b = E.log(Y), E.log1p(X * X)
    h, d = (a - b).numpy(), ingest_columns([a - b])[0]
    ok = np.isfinite(h)
    scale = np.maximum(np.abs(a.numpy()), np.abs(b.numpy()))[ok]
    assert np.array_equal(ok, np.isfinite(d)) and np.all(np.abs(h[ok] - d[ok]) <= 2 * np.spacing(scale))


def test_invalid_programs_are_rejected_not_run():
    import ctypes as C

    from gwinferno_amd import _native as N
    from gwinferno_amd import expr as E

    lib = N.load_library()
    x = np.linspace(0.0, 1.0, 64)
    out = np.empty(64)
    ptrs = (N._DP * 1)(N.as_dp(out))
    good = E.compile_program([E.log(E.Sym.src(x))])
    st, keep = N.ingest_program(good)
    assert lib.gwi_ingest_columns(C.byref(st), 64, 1, ptrs, -1) == 0 and np.allclose(out[1:], np.log(x[1:]), rtol=1e-15)
    for breakage in ("reg", "source", "nostore", "opcode", "readfirst"):
        prog = E.compile_program([E.log(E.Sym.src(x))])
        ops = list(prog.ops)
        if breakage == "reg":
            ops[1] = (ops[1][0], 63, ops[1][2], 0, 0, 0.0)          # destination register beyond n_regs
        elif breakage == "source":
            ops[0] = (E.ING_LOAD, 0, 5, 0, 0, 0.0)                  # source index out of range
        elif breakage == "nostore":
            ops = ops[:-1]                                           # column 0 never stored
        elif breakage == "readfirst":
            ops = [ops[1], ops[0]] + ops[2:]                         # the logarithm reads its register before the LOAD has written it
        else:
            ops[1] = (99, ops[1][1], ops[1][2], 0, 0, 0.0)
        prog.ops = ops
        st, keep = N.ingest_program(prog)
        assert lib.gwi_ingest_columns(C.byref(st), 64, 1, ptrs, -1) == -1, breakage  # GWI_ERR_INVALID


ALL = ["plpeak", "plpeak_full", "plpeak_smooth", "pl_test", "plpeak_default_tilt", "plpeak_iid_spins", "bspline_test", "bspline_iid", "bspline_full", "bspline_defaults",
       "bspline_misc", "bspline_chieff", "bspline_component_masses", "bspline_independent_masses", "bspline_redshift", "bspline_redshift_raw", "chm_powerlaw", "chm_bspline"]


@pytest.mark.parametrize("name", ALL)
def test_device_setup_equals_host_setup(name):
    """Every composition: the columns the ingest kernel leaves in HBM against the NumPy evaluation of the same expressions,
    and an evaluation of the engines built either way."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.lazy import INJ, PE
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(7, 300, 5000, seed=91)
    pe["mass_1"][2, :5] = [np.nan, 4.0, 150.0, 5.0, 100.0]   # NaN, outside and exactly on the truncation bounds
    pe["redshift"][3, :2] = [pe["redshift"].max(), 1e-3]
    inj["mass_ratio"][:4] = [1.0, 0.0, 1.5, np.nan]
    dev = COMPOSITIONS[name](pe, inj).engine(device_setup=True)
    host = COMPOSITIONS[name](pe, inj).engine(device_setup=False)
    assert dev.device_setup and not host.device_setup
    bm = dev.bound
    n_excluded = 0
    for side, exprs in ((PE, bm.pe_exprs), (INJ, bm.inj_exprs)):
        cols = bm.resident_columns(side)  # NumPy's columns, spline coordinates as knot coordinates (what the engine keeps)
        for c in range(len(exprs)):
            d, h = dev.read_column(side, c), host.read_column(side, c)
            assert np.array_equal(h, cols[c])                        # the host engine holds what NumPy computed
            assert np.array_equal(np.isneginf(d), np.isneginf(h)), (name, side, c)   # the same samples are excluded
            assert np.all(np.isfinite(d) | np.isneginf(d))
            if cols[c] is (bm.pe_cols if side == PE else bm.inj_cols)[c]:
                assert _ulp_close(d, h, ulps=4), (name, side, c, np.nanmax(np.abs(d - h)))
            else:  # a knot coordinate (x - lo) / dx of a logarithm: the ulp of x, not of the difference
                assert np.max(np.abs(d - h)) <= 1e-13, (name, side, c, np.max(np.abs(d - h)))
            n_excluded += int(np.isneginf(d).sum()) if c == len(exprs) - 1 else 0
    assert n_excluded > 0
    rng = np.random.default_rng(5)
    comp = COMPOSITIONS[name](pe, inj)
    for _ in range(2):
        th = dev.bound.theta_of(comp.weights(draw_params(name, rng), True))
        a = dev.evaluate(th, total, min_neff_cut=False)
        b = host.evaluate(th, total, min_neff_cut=False)
        assert abs(a.log_likelihood - b.log_likelihood) <= 1e-11 * abs(b.log_likelihood)
        assert np.allclose(a.grad, b.grad, rtol=1e-9, atol=1e-9 * np.max(np.abs(b.grad)))
        assert np.all(np.isfinite(b.log_bfs)) and np.allclose(a.log_bfs, b.log_bfs, rtol=1e-11, atol=1e-11)
    dev.close()
    host.close()


def test_float32_catalog_and_sharded_engines_are_set_up_on_the_device():
    """float32 sources go up as float32 and are widened by the kernel; a rank's engine ingests its own block of events and
    its own slice of the injections."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.engine import NativePopulationLikelihood
    from gwinferno_amd.lazy import INJ, PE
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(9, 200, 3000, seed=17)
    pe32 = {k: v.astype(np.float32) for k, v in pe.items()}
    inj32 = {k: v.astype(np.float32) for k, v in inj.items()}
    from gwinferno_amd import models as M

    z_model = M.PowerlawRedshiftModel(z_pe=pe32["redshift"], z_inj=inj32["redshift"])

    def weights(d, lamb=2.0):
        return M.plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], -2.3, 1.1, 5.0, 100.0, 33.0, 4.0, 0.1) * z_model(d["redshift"], lamb) / d["prior"]

    wp, wi = weights(pe32), weights(inj32)
    full_dev = NativePopulationLikelihood(wp, wi, z_model.normalization(2.0), device_setup=True)
    full_host = NativePopulationLikelihood(wp, wi, z_model.normalization(2.0), device_setup=False)
    for side, n_cols in ((PE, len(full_dev.bound.pe_exprs)), (INJ, len(full_dev.bound.inj_exprs))):
        for c in range(n_cols):
            assert _ulp_close(full_dev.read_column(side, c), full_host.read_column(side, c), ulps=4)
    th = full_dev.bound.theta_of(wp)
    ref = full_host.evaluate(th, total, min_neff_cut=False)
    got = full_dev.evaluate(th, total, min_neff_cut=False)
    assert abs(got.log_likelihood - ref.log_likelihood) <= 1e-11 * abs(ref.log_likelihood)
    # the same numbers from the float64 copy of the float32 catalog (what the host path converts to)
    pe64 = {k: v.astype(np.float64) for k, v in pe32.items()}
    inj64 = {k: v.astype(np.float64) for k, v in inj32.items()}
    z64 = M.PowerlawRedshiftModel(z_pe=pe64["redshift"], z_inj=inj64["redshift"])
    w64 = lambda d: M.plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], -2.3, 1.1, 5.0, 100.0, 33.0, 4.0, 0.1) * z64(d["redshift"], 2.0) / d["prior"]  # noqa: E731
    e64 = NativePopulationLikelihood(w64(pe64), w64(inj64), z64.normalization(2.0), device_setup=True)
    for c in range(len(e64.bound.pe_exprs)):
        assert np.array_equal(e64.read_column(PE, c), full_dev.read_column(PE, c))
    e64.close()
    # shards
    world = 4
    for r in range(world):
        shard = NativePopulationLikelihood(wp, wi, z_model.normalization(2.0), rank=r, world=world, device_setup=True)
        (e0, e1), (j0, j1) = shard.event_range, shard.inj_range
        for c in range(len(shard.bound.pe_exprs)):
            assert np.array_equal(shard.read_column(PE, c), full_dev.read_column(PE, c)[e0:e1])
            assert np.array_equal(shard.read_column(INJ, c), full_dev.read_column(INJ, c)[j0:j1])
        shard.close()
    full_dev.close()
    full_host.close()


def test_random_expression_graphs_on_the_device():
    """Random expression DAGs without logarithms (arithmetic, comparisons, logic, selection, sqrt, both table operations, a
    float32 source, special values in the data): the ingest kernel against the NumPy evaluation, bit for bit."""
    from gwinferno_amd import expr as E
    from gwinferno_amd.engine import ingest_columns

    rng = np.random.default_rng(21)
    n = 3001
    data = [rng.uniform(-2.0, 5.0, n), rng.lognormal(size=n), rng.uniform(0.0, 1.0, n).astype(np.float32)]
    data[0][:6] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 5.0]
    data[1][:3] = [0.0, np.inf, 1e-310]
    grid = np.linspace(-1.0, 4.0, 23)
    vals = np.cos(grid)
    unary = ["neg", "abs", "sqrt", "isfinite", "not"]
    binary = ["add", "sub", "mul", "div", "lt", "gt", "le", "ge", "and", "or"]
    for trial in range(60):
        pool = [E.Sym.src(d) for d in data] + [E.Sym.const(rng.choice([0.0, 1.0, -1.5, 3.0, np.inf]))]
        pick = lambda: pool[rng.integers(len(pool))]  # noqa: E731
        for _ in range(rng.integers(3, 30)):
            kind = rng.choice(["u", "b", "b", "w", "i", "g"])
            if kind == "u":
                pool.append(E.Sym(str(rng.choice(unary)), (pick(),)))
            elif kind == "b":
                pool.append(E.Sym(str(rng.choice(binary)), (pick(), pick())))
            elif kind == "w":
                pool.append(E.where(pick(), pick(), pick()))
            elif kind == "i":
                pool.append(E.interp(pick(), grid, vals))
            else:
                pool.append(E.gridindex(pick(), grid))
        outs = [e for e in pool[-4:] if e.shape != ()] or [pool[0]]
        host = [np.broadcast_to(np.asarray(h, dtype=np.float64), (n,)) for h in E.evaluate(outs)]
        dev = ingest_columns(outs)
        for j, (h, d) in enumerate(zip(host, dev)):
            bad = np.flatnonzero(~((h == d) | (np.isnan(h) & np.isnan(d))))
            assert bad.size == 0, (trial, j, outs[j].key[:3], bad[:5], h[bad[:5]], d[bad[:5]])


def test_log_m1_from_the_mass_spline_column(monkeypatch):
    """Config 3's model with and without the fold of the ratio term's log m1 into the m1 spline's knot-coordinate column
    (GWI_RATIO_LOGM_FROM_SPLINE): one column less in HBM, the same likelihood, sites and gradient."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(9, 500, 8000, seed=5)
    pe["mass_1"][1, :3] = [5.0, 100.0, 4.9]  # on the edges of the spline's domain and just outside
    folded = COMPOSITIONS["bspline_iid"](pe, inj).engine()
    monkeypatch.setenv("GWI_FOLD_LOGM", "0")
    plain_comp = COMPOSITIONS["bspline_iid"](pe, inj)
    plain = plain_comp.engine()
    assert len(folded.bound.pe_exprs) == 8 and len(plain.bound.pe_exprs) == 9 and folded.bytes_per_sample == 64
    rng = np.random.default_rng(12)
    for _ in range(3):
        th = plain.bound.theta_of(plain_comp.weights(draw_params("bspline_iid", rng), True))
        a, b = folded.evaluate(th, total, min_neff_cut=False), plain.evaluate(th, total, min_neff_cut=False)
        assert abs(a.log_likelihood - b.log_likelihood) <= 1e-12 * abs(b.log_likelihood)
        assert np.allclose(a.log_bfs, b.log_bfs, rtol=0, atol=1e-11) and np.allclose(a.grad, b.grad, rtol=1e-10, atol=1e-10 * np.max(np.abs(b.grad)))
    folded.close()
    plain.close()


def test_oversized_setup_program_falls_back_to_the_host_path(monkeypatch):
    """A setup program beyond the ingest kernel's register file: the default (auto) setup computes the columns on the host
    instead, with a warning, and gives the device path's results; an explicit device_setup=True raises."""
    import warnings

    from gwinferno_amd import expr as E
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    monkeypatch.delenv("GWI_HOST_SETUP", raising=False)  # (the suite is also run with the host path forced: this test is about the default)
    pe, inj, total = make_catalog(5, 200, 3000, seed=17)
    ref = COMPOSITIONS["bspline_test"](pe, inj)
    eng_ref = ref.engine()
    th = ref.theta(draw_params("bspline_test", np.random.default_rng(2)))
    want = eng_ref.evaluate(th, total, min_neff_cut=False)
    assert eng_ref.device_setup
    monkeypatch.setattr(E, "MAX_REGS", 4)
    with pytest.raises(ValueError):
        COMPOSITIONS["bspline_test"](pe, inj).engine(device_setup=True)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        eng = COMPOSITIONS["bspline_test"](pe, inj).engine()
    assert not eng.device_setup and any("computing the catalog columns on the host" in str(w.message) for w in caught)
    got = eng.evaluate(th, total, min_neff_cut=False)
    assert abs(got.log_likelihood - want.log_likelihood) <= 1e-12 * abs(want.log_likelihood)
    eng.close()
    eng_ref.close()
