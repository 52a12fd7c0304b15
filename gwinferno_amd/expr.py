"""Per-sample setup expressions: ONE description of the reference's one-time, theta-independent work, two evaluators.

The reference does its per-sample setup eagerly in NumPy/JAX when a model object is constructed: validity masks
(models/bsplines/single.py:54-55, distributions.py:119,143,162, parametric.py:141-145), coordinate transforms
(``jnp.log`` of the mass / redshift columns, interpolation.py:357,447), ``dVc/dz`` per sample by linear interpolation
into the comoving-distance table (cosmology.py:95-120), ``1 / prior`` (examples/simple_bspline_example.py:58-71).  Here
those operations are recorded as small expression graphs (:class:`Sym`) over the caller's arrays -- building one costs
nothing per sample -- and evaluated once per engine, either

* on the device: :func:`compile_program` flattens the graphs of every column and of ``kappa`` into the register
  program of ``include/gwi_engine.h`` (``gwi_ingest_program``), which ``gwi_create_ingest`` runs in one HIP kernel over
  the raw catalog columns it uploads (the setup path), or
* on the host: :meth:`Sym.numpy` walks the same graph with NumPy (engines without a device, the oracles, the parity
  test of the device kernel).

Both evaluate the SAME operation sequence in fp64 without fused multiply-adds, so every arithmetic step, comparison,
selection, square root and table interpolation agrees to the bit; only ``log`` / ``log1p`` come from different maths
libraries (<= 1 ulp each).
"""
import numpy as np

# ---- opcodes: keep in step with include/gwi_engine.h (GWI_ING_*) ---------------------------------------------------
ING_LOAD, ING_CONST = 0, 1
ING_LOG, ING_LOG1P, ING_NEG, ING_ABS, ING_NOT, ING_SQRT, ING_ISFINITE = 2, 3, 4, 5, 6, 7, 8
ING_ADD, ING_SUB, ING_MUL, ING_DIV, ING_LT, ING_GT, ING_LE, ING_GE, ING_AND, ING_OR = 10, 11, 12, 13, 14, 15, 16, 17, 18, 19
ING_WHERE, ING_INTERP, ING_GRIDINDEX, ING_STORE = 20, 21, 22, 23

_UNARY = {"log": ING_LOG, "log1p": ING_LOG1P, "neg": ING_NEG, "abs": ING_ABS, "not": ING_NOT, "sqrt": ING_SQRT, "isfinite": ING_ISFINITE}
_BINARY = {"add": ING_ADD, "sub": ING_SUB, "mul": ING_MUL, "div": ING_DIV, "lt": ING_LT, "gt": ING_GT, "le": ING_LE, "ge": ING_GE, "and": ING_AND,
           "or": ING_OR}
_BOOLEAN = {"not", "isfinite", "lt", "gt", "le", "ge", "and", "or"}

MAX_REGS = 64  # GWI_INGEST_MAX_REGS


class Sym:
    """Node of a per-sample expression.  Leaves: ``src`` (a caller's array, keyed by identity -- float64 or float32,
    anything else is converted once), ``const`` (a Python float).  ``interp`` / ``gridindex`` carry their tables in ``k``."""

    __slots__ = ("op", "args", "k", "key", "shape")

    def __init__(self, op, args=(), k=None):
        self.op, self.args, self.k = op, tuple(args), k
        if op == "src":
            self.key = ("src", id(k))
            self.shape = np.shape(k)
        elif op == "const":
            self.key = ("const", float(k).hex())
            self.shape = ()
        else:
            tab = tuple(id(t) for t in k) if k is not None else ()
            self.key = (op,) + tuple(a.key for a in self.args) + tab
            shapes = [a.shape for a in self.args if a.shape != ()]
            self.shape = shapes[0] if shapes else ()
            if any(s != self.shape for s in shapes):
                raise ValueError(f"per-sample arrays of different shapes in one expression: {sorted(set(shapes))}")

    # ---- construction ---------------------------------------------------------------------------------------------
    @staticmethod
    def src(array):
        return array if isinstance(array, Sym) else Sym("src", (), array)

    @staticmethod
    def const(value):
        return Sym("const", (), float(value))

    @staticmethod
    def of(x):
        """Sym of a Sym, a per-sample array or a plain number."""
        if isinstance(x, Sym):
            return x
        return Sym.const(x) if np.ndim(x) == 0 else Sym.src(x)

    def _bin(self, name, other, swap=False):
        o = Sym.of(other)
        return Sym(name, (o, self) if swap else (self, o))

    def __add__(self, o): return self._bin("add", o)            # noqa: E704
    def __radd__(self, o): return self._bin("add", o, True)     # noqa: E704
    def __sub__(self, o): return self._bin("sub", o)            # noqa: E704
    def __rsub__(self, o): return self._bin("sub", o, True)     # noqa: E704
    def __mul__(self, o): return self._bin("mul", o)            # noqa: E704
    def __rmul__(self, o): return self._bin("mul", o, True)     # noqa: E704
    def __truediv__(self, o): return self._bin("div", o)        # noqa: E704
    def __rtruediv__(self, o): return self._bin("div", o, True)  # noqa: E704
    def __lt__(self, o): return self._bin("lt", o)              # noqa: E704
    def __gt__(self, o): return self._bin("gt", o)              # noqa: E704
    def __le__(self, o): return self._bin("le", o)              # noqa: E704
    def __ge__(self, o): return self._bin("ge", o)              # noqa: E704
    def __and__(self, o): return self._bin("and", o)            # noqa: E704
    def __or__(self, o): return self._bin("or", o)              # noqa: E704
    def __neg__(self): return Sym("neg", (self,))               # noqa: E704
    def __abs__(self): return Sym("abs", (self,))               # noqa: E704
    def __invert__(self): return Sym("not", (self,))            # noqa: E704
    __hash__ = None  # comparisons build expressions: a Sym must never be used as a dictionary key (use .key)

    def __bool__(self):
        raise TypeError("a per-sample expression has no truth value (use & | ~)")

    # ---- inspection -----------------------------------------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    def sources(self):
        """The distinct source arrays of the expression, in first-use order."""
        out, seen, stack = [], set(), [self]
        while stack:
            n = stack.pop()
            if n.op == "src":
                if id(n.k) not in seen:
                    seen.add(id(n.k))
                    out.append(n.k)
            stack.extend(reversed(n.args))
        return out

    def substitute(self, fn, memo=None):
        """The same expression over other arrays: every source ``a`` becomes ``fn(a)`` (one call per distinct array)."""
        memo = {} if memo is None else memo
        hit = memo.get(self.key)
        if hit is not None:
            return hit
        if self.op == "src":
            new = Sym.src(fn(self.k))
        elif self.op == "const":
            new = self
        else:
            new = Sym(self.op, [a.substitute(fn, memo) for a in self.args], self.k)
        memo[self.key] = new
        return new

    # ---- host evaluation ------------------------------------------------------------------------------------------
    def numpy(self):
        """Evaluate with NumPy (fp64; booleans as bool arrays; a scalar where no source array enters)."""
        return evaluate([self])[0]


def _post_order(outputs):
    """Distinct nodes of the graphs below ``outputs`` in evaluation order + {key: index}."""
    nodes, number = [], {}
    for out in outputs:
        stack = [(out, False)]
        while stack:  # iterative: expression chains (kappa) can be long
            n, done = stack.pop()
            if n.key in number:
                continue
            if done:
                number[n.key] = len(nodes)
                nodes.append(n)
                continue
            stack.append((n, True))
            stack.extend((a, False) for a in reversed(n.args) if a.key not in number)
    return nodes, number


def evaluate(outputs):
    """NumPy values of several expressions at once: common sub-expressions are computed once and every intermediate is
    dropped after its last use (a catalog-sized temporary per node would otherwise stay alive until the end)."""
    nodes, number = _post_order(outputs)
    last = list(range(len(nodes)))
    for i, n in enumerate(nodes):
        for a in n.args:
            last[number[a.key]] = i
    keep = {number[o.key] for o in outputs}
    val = [None] * len(nodes)
    with np.errstate(all="ignore"):
        for i, n in enumerate(nodes):
            val[i] = _numpy_node(n, [val[number[a.key]] for a in n.args])
            for a in n.args:
                j = number[a.key]
                if last[j] == i and j not in keep:
                    val[j] = None
    return [val[number[o.key]] for o in outputs]


def _numpy_node(n, v):
    op = n.op
    if op == "src":
        return np.asarray(n.k, dtype=np.float64)
    if op == "const":
        return n.k
    if op in ("log", "log1p", "neg", "abs", "sqrt", "isfinite"):
        x = np.asarray(v[0], dtype=np.float64)  # booleans count as 0.0 / 1.0, as in the device registers (NumPy would take their logarithm in float16)
        if op == "log":
            return np.log(x)
        if op == "log1p":
            return np.log1p(x)
        if op == "neg":
            return -x
        if op == "abs":
            return np.abs(x)
        if op == "sqrt":
            return np.sqrt(x)
        return np.isfinite(x)
    if op == "not":
        return ~_as_bool(v[0])
    if op in ("and", "or"):
        a, b = _as_bool(v[0]), _as_bool(v[1])
        return (a & b) if op == "and" else (a | b)
    if op in ("add", "sub", "mul", "div"):
        a, b = (np.asarray(x, dtype=np.float64) for x in v)
        return a + b if op == "add" else a - b if op == "sub" else a * b if op == "mul" else a / b
    if op in ("lt", "gt", "le", "ge"):
        a, b = (np.asarray(x, dtype=np.float64) for x in v)
        return a < b if op == "lt" else a > b if op == "gt" else a <= b if op == "le" else a >= b
    if op == "where":
        return np.where(_as_bool(v[0]), np.asarray(v[1], dtype=np.float64), np.asarray(v[2], dtype=np.float64))
    if op == "interp":
        return np.interp(np.asarray(v[0], dtype=np.float64), n.k[0], n.k[1])
    if op == "gridindex":
        # fractional index j + f of x in the grid, exactly the piece and weight np.interp uses (end values held outside
        # the grid); NaN stays NaN (excluded at bind)
        g = n.k[0]
        x = np.asarray(v[0], dtype=np.float64)
        j = np.clip(np.searchsorted(g, x, side="right") - 1, 0, g.size - 2)
        f = np.clip((x - g[j]) / (g[j + 1] - g[j]), 0.0, 1.0)
        return j + f
    raise ValueError(f"unknown expression node {op!r}")


def _as_bool(x):
    x = np.asarray(x)
    return x if x.dtype == np.bool_ else x != 0


# ---- free functions (NumPy-like spelling inside the model code) ------------------------------------------------------
def log(x): return Sym("log", (Sym.of(x),))            # noqa: E704
def log1p(x): return Sym("log1p", (Sym.of(x),))        # noqa: E704
def sqrt(x): return Sym("sqrt", (Sym.of(x),))          # noqa: E704
def isfinite(x): return Sym("isfinite", (Sym.of(x),))  # noqa: E704


def where(c, a, b):
    return Sym("where", (Sym.of(c), Sym.of(a), Sym.of(b)))


def interp(x, xp, fp):
    """``np.interp(x, xp, fp)`` (end values held outside the table, as the reference's ``jnp.interp``)."""
    xp, fp = np.ascontiguousarray(xp, dtype=np.float64), np.ascontiguousarray(fp, dtype=np.float64)
    if xp.ndim != 1 or xp.shape != fp.shape or xp.size < 2:
        raise ValueError("interp tables must be two 1-D arrays of one length >= 2")
    return Sym("interp", (Sym.of(x),), (xp, fp))


def gridindex(x, grid):
    return Sym("gridindex", (Sym.of(x),), (np.ascontiguousarray(grid, dtype=np.float64),))


# ---- compilation ------------------------------------------------------------------------------------------------------
class Program:
    """Flat register program for one side: ``ops`` rows ``(op, dst, a, b, c, k)``; ``sources`` the distinct source arrays
    (``ING_LOAD``'s ``a`` indexes them); ``tables`` the interpolation tables (``b`` / ``c`` of ``ING_INTERP``, ``b`` of
    ``ING_GRIDINDEX``); ``n_regs`` registers; ``ING_STORE`` writes register ``a`` to output column ``dst``."""

    def __init__(self):
        self.ops, self.sources, self.tables, self.n_regs, self.n_out = [], [], [], 0, 0


def compile_program(outputs):
    """``outputs``: one Sym per output column (the last one is kappa by the engine's convention).  Common sub-expressions
    are evaluated once; registers are reused after a value's last use."""
    prog = Program()
    prog.n_out = len(outputs)
    src_index, tab_index = {}, {}

    def table(t):
        if id(t) not in tab_index:
            tab_index[id(t)] = len(prog.tables)
            prog.tables.append(t)
        return tab_index[id(t)]

    # 1. value numbering in post-order over all outputs
    nodes, number = _post_order(outputs)
    # 2. last use of every value (stores count as uses at the very end of the program: they are emitted last, so that a
    #    store's register is still live)
    last = [i for i in range(len(nodes))]
    for i, n in enumerate(nodes):
        for a in n.args:
            last[number[a.key]] = max(last[number[a.key]], i)
    for out in outputs:
        last[number[out.key]] = len(nodes) + 1
    # 3. emit with linear-scan register reuse
    free, reg_of, next_reg = [], {}, 0
    for i, n in enumerate(nodes):
        a = [reg_of[number[x.key]] for x in n.args]
        # arguments whose last use is this node give their register back BEFORE the destination is chosen: an op may
        # overwrite one of its own inputs (the kernel reads all inputs first)
        for x in n.args:
            j = number[x.key]
            if last[j] == i and reg_of[j] not in free:
                free.append(reg_of[j])
        if free:
            dst = free.pop()
        else:
            dst, next_reg = next_reg, next_reg + 1
        reg_of[i] = dst
        if n.op == "src":
            if id(n.k) not in src_index:
                src_index[id(n.k)] = len(prog.sources)
                prog.sources.append(n.k)
            prog.ops.append((ING_LOAD, dst, src_index[id(n.k)], 0, 0, 0.0))
        elif n.op == "const":
            prog.ops.append((ING_CONST, dst, 0, 0, 0, float(n.k)))
        elif n.op in _UNARY:
            prog.ops.append((_UNARY[n.op], dst, a[0], 0, 0, 0.0))
        elif n.op in _BINARY:
            prog.ops.append((_BINARY[n.op], dst, a[0], a[1], 0, 0.0))
        elif n.op == "where":
            prog.ops.append((ING_WHERE, dst, a[0], a[1], a[2], 0.0))
        elif n.op == "interp":
            prog.ops.append((ING_INTERP, dst, a[0], table(n.k[0]), table(n.k[1]), 0.0))
        elif n.op == "gridindex":
            prog.ops.append((ING_GRIDINDEX, dst, a[0], table(n.k[0]), 0, 0.0))
        else:
            raise ValueError(f"unknown expression node {n.op!r}")
        if last[i] == i and dst not in free:  # a value nobody reads (cannot happen for reachable nodes; kept for safety)
            free.append(dst)
    for c, out in enumerate(outputs):
        prog.ops.append((ING_STORE, c, reg_of[number[out.key]], 0, 0, 0.0))
    prog.n_regs = next_reg
    if prog.n_regs > MAX_REGS:
        raise ValueError(f"setup program needs {prog.n_regs} registers; the device evaluator has {MAX_REGS}")
    return prog


def run_program_numpy(prog, n):
    """Interpret a compiled program with NumPy (tests: the compiler against :meth:`Sym.numpy`)."""
    regs = [None] * max(prog.n_regs, 1)
    out = [None] * prog.n_out
    with np.errstate(all="ignore"):
        for op, dst, a, b, c, k in prog.ops:
            if op == ING_LOAD:
                regs[dst] = np.asarray(prog.sources[a], dtype=np.float64).reshape(-1)
            elif op == ING_CONST:
                regs[dst] = np.full(n, k)
            elif op == ING_STORE:
                out[dst] = np.array(regs[a], dtype=np.float64)
            elif op == ING_WHERE:
                regs[dst] = np.where(regs[a] != 0, regs[b], regs[c])
            elif op == ING_INTERP:
                regs[dst] = np.interp(regs[a], prog.tables[b], prog.tables[c])
            elif op == ING_GRIDINDEX:
                regs[dst] = _numpy_node(Sym("gridindex", (Sym.const(0.0),), (prog.tables[b],)), [regs[a]])
            else:
                name = {v: k_ for k_, v in {**_UNARY, **_BINARY}.items()}[op]
                args = [regs[a]] if name in _UNARY else [regs[a], regs[b]]
                r = _numpy_node(Sym(name, [Sym.const(0.0)] * len(args)), args)
                regs[dst] = r.astype(np.float64) if name in _BOOLEAN else r
    return out
