"""A MINIMAL stand-in for the parts of JAX that gwinferno_amd's custom_vjp adapter touches (likelihood._evaluate_jax):
``custom_vjp`` / ``defvjp``, ``pure_callback``, ``ShapeDtypeStruct``, ``jax.config`` (the x64 switch), ``jax.dtypes`` and a
handful of ``jax.numpy`` functions over an opaque ``Tracer`` value -- plus toy versions of the transformations NUTS applies
to a potential: ``value_and_grad`` (a reverse sweep over the recorded operations, through the custom_vjp's own backward
rule), ``vmap`` (a leading batch axis that every operation keeps in front; ``pure_callback`` then makes ONE call with the
batch, as ``vmap_method="broadcast_all"`` does) and ``jit`` (identity).  TEST INFRASTRUCTURE for
tests/test_jax_adapter_cpu.py only (JAX is not installable in the build or test images): it checks the adapter's plumbing --
dtypes and shapes declared to pure_callback, primal / forward agreement, what the backward rule returns and where it ends
up -- not JAX itself."""
import numpy as _np

from . import numpy  # noqa: F401  (jax.numpy)
from .numpy import X64, Tracer, _unwrap, _wrap, canonicalize_dtype


class _Config:
    @property
    def jax_enable_x64(self):
        return X64[0]

    def update(self, name, value):
        if name != "jax_enable_x64":
            raise AttributeError(name)
        X64[0] = bool(value)

    def read(self, name):
        if name != "jax_enable_x64":
            raise AttributeError(name)
        return X64[0]


config = _Config()


class dtypes:  # jax.dtypes
    canonicalize_dtype = staticmethod(canonicalize_dtype)


class ShapeDtypeStruct:
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), _np.dtype(dtype)


def _check(result, spec, where, batch=None):
    arr = _np.asarray(result)
    want_shape = spec.shape if batch is None else (batch,) + spec.shape
    want_dtype = canonicalize_dtype(spec.dtype)  # with x64 off a declared float64 MEANS float32, and the host function must return that
    if arr.shape != want_shape:
        raise TypeError(f"pure_callback {where}: host function returned shape {arr.shape}, declared {want_shape}")
    if arr.dtype != want_dtype:
        raise TypeError(f"pure_callback {where}: host function returned dtype {arr.dtype}, expected {want_dtype} (declared {spec.dtype}, x64 {'on' if X64[0] else 'off'})")
    return _wrap(arr, batched=batch is not None)


CALLBACK_CALLS = []


def pure_callback(host, result_shape_dtypes, *args, vmap_method=None):
    """Calls ``host`` with concrete NumPy arrays and verifies the result against the declared structure.  Batched arguments
    (inside the shim's ``vmap``): ONE call with the batch axis in front of every argument when ``vmap_method ==
    "broadcast_all"``, one call per example otherwise (what ``vmap_method="sequential"`` / the legacy default amount to)."""
    batch = next((a.val.shape[0] for a in args if isinstance(a, Tracer) and a.batched), None)
    many = isinstance(result_shape_dtypes, (tuple, list))
    specs = list(result_shape_dtypes) if many else [result_shape_dtypes]

    def call(vals, b):
        out = host(*vals)
        CALLBACK_CALLS.append(host)
        outs = list(out) if many else [out]
        if many and (not isinstance(out, (tuple, list)) or len(out) != len(specs)):
            raise TypeError("pure_callback: result structure differs from the declared one")
        return [_check(o, s, f"output {i}", b) for i, (o, s) in enumerate(zip(outs, specs))]

    if batch is None:
        res = call([_unwrap(a) for a in args], None)
    elif vmap_method == "broadcast_all":
        vals = [a.val if (isinstance(a, Tracer) and a.batched) else _np.broadcast_to(_np.asarray(_unwrap(a)), (batch,) + _np.shape(_unwrap(a))) for a in args]
        res = call(vals, batch)
    else:
        per = [call([(a.val[i] if (isinstance(a, Tracer) and a.batched) else _unwrap(a)) for a in args], None) for i in range(batch)]
        res = [_wrap(_np.stack([p[j].val for p in per]), batched=True) for j in range(len(specs))]
    return tuple(res) if many else res[0]


class _VjpNode:
    """The joint node of one custom_vjp call on the tape: gathers the cotangents of all its outputs, then applies the user's
    backward rule once (per example under vmap, as a batched backward rule would)."""

    def __init__(self, owner, residuals, outs, args):
        self.owner, self.residuals, self.outs, self.args = owner, residuals, outs, args
        self.batched = any(isinstance(o, Tracer) and o.batched for o in outs)

    def pull(self, slots):
        cts = tuple(slots.get(i, _np.zeros_like(o.val)) for i, o in enumerate(self.outs))
        single = len(self.outs) == 1 and not self.owner.tuple_out
        if not self.batched:
            grads = self.owner.bwd(self.residuals, cts[0] if single else cts)
        else:
            n = self.outs[0].val.shape[0]
            pick = lambda tree, i: tuple(pick(t, i) for t in tree) if isinstance(tree, tuple) else (_unwrap(tree)[i] if getattr(tree, "batched", False) else _unwrap(tree))  # noqa: E731
            one = lambda tree, i: tuple(one(t, i) for t in tree) if isinstance(tree, tuple) else _np.asarray(tree)[i]  # noqa: E731  (cotangents: plain arrays, batch axis in front)
            per = [self.owner.bwd(pick(self.residuals, i), one(cts[0] if single else cts, i)) for i in range(n)]
            grads = tuple(_np.stack([_unwrap(p[j]) for p in per]) for j in range(len(per[0])))
        if not isinstance(grads, tuple) or len(grads) != len(self.args):
            raise TypeError("custom_vjp: the backward rule must return one cotangent per primal argument")
        return [(a, _unwrap(g)) for a, g in zip(self.args, grads) if isinstance(a, Tracer)]


class custom_vjp:
    """Records the forward / backward rules.  Calling the function runs BOTH the primal body and the forward rule (they
    must agree, as JAX requires), keeps the residuals so that a test can pull cotangents back with :meth:`pull`, and puts the
    call on the tape so that :func:`value_and_grad` reaches the arguments through the backward rule."""

    last = None

    def __init__(self, fun):
        self.fun, self.fwd, self.bwd, self.tuple_out = fun, None, None, True

    def defvjp(self, fwd, bwd):
        self.fwd, self.bwd = fwd, bwd

    def __call__(self, *args):
        if self.fwd is None:
            raise RuntimeError("custom_vjp called before defvjp")
        primal = self.fun(*args)
        out, residuals = self.fwd(*args)
        self.tuple_out = isinstance(out, tuple)
        flat_p = primal if isinstance(primal, tuple) else (primal,)
        flat_o = out if isinstance(out, tuple) else (out,)
        if len(flat_p) != len(flat_o) or any(not _np.array_equal(_unwrap(a), _unwrap(b), equal_nan=True) for a, b in zip(flat_p, flat_o)):
            raise AssertionError("custom_vjp: the forward rule's outputs differ from the primal function's")
        custom_vjp.last = (self, residuals, out, args)
        node = _VjpNode(self, residuals, flat_o, args)
        taped = tuple(Tracer(o.val, [(node, (lambda ct, i=i: (i, ct)))], o.batched) for i, o in enumerate(flat_o))
        return taped if self.tuple_out else taped[0]

    @classmethod
    def pull(cls, cotangents):
        """Apply the recorded backward rule of the most recent call to ``cotangents`` (same structure as the outputs)."""
        self, residuals, out, args = cls.last
        grads = self.bwd(residuals, cotangents)
        if not isinstance(grads, tuple) or len(grads) != len(args):
            raise TypeError("custom_vjp: the backward rule must return one cotangent per primal argument")
        return tuple(_unwrap(g) for g in grads)


def _backprop(out, seed):
    """Reverse sweep from ``out`` (a Tracer) with cotangent ``seed``; returns {id(leaf tracer): cotangent}."""
    order, seen = [], set()

    def visit(n):
        if id(n) in seen:
            return
        seen.add(id(n))
        for p, _ in (n.parents if isinstance(n, Tracer) else [(a, None) for a in n.args if isinstance(a, Tracer)]):
            visit(p)
        order.append(n)

    visit(out)
    acc = {id(out): seed}
    for n in reversed(order):
        ct = acc.get(id(n))
        if ct is None:
            continue
        if isinstance(n, _VjpNode):
            pairs = n.pull(ct)
        else:
            pairs = [(p, f(ct)) for p, f in n.parents]
        for p, c in pairs:
            if isinstance(p, _VjpNode):
                i, c = c
                acc.setdefault(id(p), {})
                acc[id(p)][i] = acc[id(p)][i] + c if i in acc[id(p)] else c
            else:
                acc[id(p)] = acc[id(p)] + c if id(p) in acc else c
    return acc


def _tree_map(f, tree):
    if isinstance(tree, dict):
        return {k: _tree_map(f, v) for k, v in tree.items()}
    if isinstance(tree, (tuple, list)):
        return type(tree)(_tree_map(f, v) for v in tree)
    return f(tree)


def value_and_grad(fun, argnums=0):
    if argnums != 0:
        raise NotImplementedError
    def wrapped(first, *rest):  # noqa: E306
        inside = []

        def leaf(x):
            if isinstance(x, Tracer):  # already traced (vmap outside): a fresh leaf on the same values
                inside.append(True)
                return Tracer(x.val, [], x.batched)
            v = _np.asarray(x)
            return Tracer(v.astype(canonicalize_dtype(v.dtype)) if v.dtype in (_np.float64, _np.int64) else v)

        leaves = _tree_map(leaf, first)
        out = fun(leaves, *rest)
        if not isinstance(out, Tracer) or out.ndim != 0:
            raise TypeError("value_and_grad: the function must return a traced scalar")
        acc = _backprop(out, _np.ones_like(out.val))
        grads = _tree_map(lambda t: _np.asarray(acc.get(id(t), _np.zeros_like(t.val)), dtype=t.val.dtype), leaves)
        if inside:
            return Tracer(out.val, [], out.batched), _tree_map(lambda g: Tracer(g, [], out.batched), grads)
        return out.val, grads

    return wrapped


def vmap(fun):
    def wrapped(*args):
        out = fun(*[_tree_map(lambda x: Tracer(_np.asarray(x), [], True), a) for a in args])
        return _tree_map(lambda t: t.val if isinstance(t, Tracer) else t, out)

    return wrapped


def jit(fun=None, **_):
    return fun if fun is not None else (lambda f: f)


__all__ = ["ShapeDtypeStruct", "pure_callback", "custom_vjp", "jit", "value_and_grad", "vmap", "config", "dtypes", "Tracer"]
