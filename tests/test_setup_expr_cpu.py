"""CPU: the setup expressions (gwinferno_amd/expr.py) -- the NumPy evaluation against hand-written NumPy, the compiled
register program (what gwi_create_ingest runs on the device) interpreted on the host against the graph evaluation for
every composition, and the structural properties the device evaluator relies on."""
import numpy as np
import pytest

from gwinferno_amd import expr as E


def test_expressions_evaluate_like_the_numpy_they_replace():
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 4, (5, 40))
    y = rng.lognormal(size=(5, 40))
    x[0, :4] = [np.nan, np.inf, 0.0, -1.0]
    X, Y = E.Sym.src(x), E.Sym.src(y)
    with np.errstate(all="ignore"):
        assert np.array_equal((~((X < 0.5) | (X > 2.5))).numpy(), ~((x < 0.5) | (x > 2.5)))
        assert np.array_equal(E.where((X <= 1) & (Y >= 1), E.log(X), -np.inf).numpy(), np.where((x <= 1) & (y >= 1), np.log(x), -np.inf), equal_nan=True)
        assert np.array_equal((X * Y - 0.3).numpy(), x * y - 0.3, equal_nan=True)
        assert np.array_equal((5.0 / Y).numpy(), 5.0 / y) and np.array_equal((1.0 - X / 3.0).numpy(), 1.0 - x / 3.0, equal_nan=True)
        g = np.linspace(-0.5, 3.5, 17)
        assert np.array_equal(E.interp(X, g, g**2).numpy(), np.interp(x, g, g**2), equal_nan=True)
    assert (X + 1.0).shape == (5, 40) and E.Sym.const(2.0).shape == () and (X + 1.0).ndim == 2
    with pytest.raises(ValueError):
        X + E.Sym.src(np.zeros(7))        # PE and injection arrays in one expression
    with pytest.raises(TypeError):
        bool(X < 1)                        # `and` / `or` / `if` on per-sample expressions are bugs
    # identity of the caller's arrays keys an expression; constants by value
    assert (E.log(X) - 1.5).key == (E.log(E.Sym.src(x)) - 1.5).key != (E.log(E.Sym.src(x.copy())) - 1.5).key


def test_compiled_programs_share_subexpressions_and_reuse_registers():
    x = np.linspace(0.1, 2.0, 50)
    X = E.Sym.src(x)
    lx = E.log(X)
    prog = E.compile_program([lx + 1.0, lx * lx, E.where(X > 1.0, lx, 0.0)])
    assert sum(op[0] == E.ING_LOG for op in prog.ops) == 1 and sum(op[0] == E.ING_LOAD for op in prog.ops) == 1
    assert len(prog.sources) == 1 and prog.n_out == 3 and [op[1] for op in prog.ops if op[0] == E.ING_STORE] == [0, 1, 2]
    chain = X
    for k in range(200):  # a long dependent chain needs two registers, not two hundred
        chain = chain * 1.0001 + float(k)
    prog = E.compile_program([chain])
    assert prog.n_regs <= 3
    out = E.run_program_numpy(prog, x.size)[0]
    assert np.array_equal(out, chain.numpy())
    wide = [E.log(X + float(k)) for k in range(E.MAX_REGS + 8)]
    total = wide[0]
    for w in wide[1:]:
        total = total + w
    assert E.compile_program([total]).n_regs <= 4     # evaluated depth-first: operands die as soon as they are summed
    with pytest.raises(ValueError):                    # every output stays live to the end: this one cannot fit
        E.compile_program(wide)


ALL = ["plpeak", "plpeak_full", "plpeak_smooth", "pl_test", "plpeak_default_tilt", "plpeak_iid_spins", "bspline_test", "bspline_iid", "bspline_full", "bspline_defaults",
       "bspline_misc", "bspline_chieff", "bspline_component_masses", "bspline_independent_masses", "bspline_redshift", "bspline_redshift_raw", "chm_powerlaw", "chm_bspline"]


@pytest.mark.parametrize("name", ALL)
def test_setup_program_of_every_composition(name):
    """What the device evaluates, interpreted op by op on the host, equals the graph evaluation bit for bit; sources are the
    caller's own arrays (nothing is copied or transformed before the upload); limits of the device evaluator hold."""
    from gwinferno_amd import _native as N
    from gwinferno_amd.compositions import COMPOSITIONS
    from gwinferno_amd.engine import bind
    from gwinferno_amd.lazy import INJ, PE
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, _ = make_catalog(4, 60, 500, seed=23)
    pe["mass_1"][1, :3] = [np.nan, 4.0, 150.0]
    comp = COMPOSITIONS[name](pe, inj)
    p = comp.placeholder()
    bm = bind(comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p))
    raw = {id(v) for d in (comp.pe, comp.inj) for v in d.values()}
    for side, cols, n in ((PE, bm.pe_cols, pe["mass_1"].size), (INJ, bm.inj_cols, inj["mass_1"].size)):
        prog = bm.program(side)
        assert prog.n_regs <= N.GWI_INGEST_MAX_REGS and len(prog.sources) <= N.GWI_INGEST_MAX_SOURCES and len(prog.tables) <= N.GWI_INGEST_MAX_TABLES
        if not name.startswith("bspline_redshift"):  # BSplineRedshift is HANDED dVc/dz arrays by its caller (single.py:398): sources too
            assert all(id(s) in raw for s in prog.sources), "a source of the setup program is not one of the caller's arrays"
        out = E.run_program_numpy(prog, n)
        assert len(out) == len(cols)
        for o, c in zip(out, cols):
            assert np.array_equal(o, c.ravel(), equal_nan=True)
        kap = cols[-1]
        assert np.all(np.isfinite(kap) | np.isneginf(kap)) and all(np.all(np.isfinite(c)) for c in cols[:-1])
        st, keep = N.ingest_program(prog)  # marshals without a device
        assert st.n_ops == len(prog.ops) and st.n_sources == len(prog.sources)
    sh = bm.program(PE, events=(1, 3))
    assert all(np.shape(s) == (2, 60) for s in sh.sources)


def test_random_expression_graphs_compile_to_equivalent_programs():
    """Property test: random expression DAGs (shared sub-expressions, every operation, special values in the data) -- the
    compiled register program, interpreted op by op, reproduces the graph evaluation bit for bit, stays within the device
    evaluator's register file, and never stores a column twice."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    rng = np.random.default_rng(5)
    n = 257
    data = [rng.uniform(-2.0, 5.0, n), rng.lognormal(size=n), rng.uniform(0.0, 1.0, n).astype(np.float32)]
    data[0][:6] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 5.0]
    data[1][:3] = [0.0, np.inf, 1e-310]
    grid = np.linspace(-1.0, 4.0, 23)
    vals = np.cos(grid)
    unary = ["log", "log1p", "neg", "abs", "sqrt", "isfinite", "not"]
    binary = ["add", "sub", "mul", "div", "lt", "gt", "le", "ge", "and", "or"]

    @st.composite
    def graphs(draw):
        pool = [E.Sym.src(d) for d in data] + [E.Sym.const(draw(st.sampled_from([0.0, 1.0, -1.5, 3.0, np.inf])))]
        for _ in range(draw(st.integers(3, 25))):
            kind = draw(st.sampled_from(["u", "b", "b", "w", "i", "g"]))
            pick = lambda: pool[draw(st.integers(0, len(pool) - 1))]  # noqa: E731
            if kind == "u":
                pool.append(E.Sym(draw(st.sampled_from(unary)), (pick(),)))
            elif kind == "b":
                pool.append(E.Sym(draw(st.sampled_from(binary)), (pick(), pick())))
            elif kind == "w":
                pool.append(E.where(pick(), pick(), pick()))
            elif kind == "i":
                pool.append(E.interp(pick(), grid, vals))
            else:
                pool.append(E.gridindex(pick(), grid))
        k = draw(st.integers(1, 4))
        return [pool[-1 - j] for j in range(k) if pool[-1 - j].shape != ()] or [pool[0]]

    @settings(max_examples=150, deadline=None)
    @given(graphs())
    def check(outs):
        want = E.evaluate(outs)
        prog = E.compile_program(outs)
        got = E.run_program_numpy(prog, n)
        assert prog.n_regs <= E.MAX_REGS and sorted(op[1] for op in prog.ops if op[0] == E.ING_STORE) == list(range(len(outs)))
        for w, g in zip(want, got):
            w = np.broadcast_to(np.asarray(w, dtype=np.float64), (n,))
            assert np.array_equal(w, g, equal_nan=True)

    check()
