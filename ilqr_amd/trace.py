"""Device dynamics derived from a System's own numpy methods.

The reference solver takes ANY object with the ``System`` duck type
(``systems.py:4-24``) and calls its ``model_f`` / ``compute_df_dx`` /
``compute_df_du`` in Python at every time step (``ilqr.py:62``, ``:87-88``).
Models written the way the reference writes its own (``systems.py:33-57``:
scalars picked out of ``x`` and ``u``, arithmetic and numpy ufuncs on them, the
results assembled with ``np.array`` / ``np.reshape``) are straight-line
arithmetic, so calling each method ONCE with symbolic scalars records the whole
computation.  This module does that, checks the record against the real methods
at random points, and writes it out as the device snippets that
``ilqr_amd/jit.py`` compiles.  Jacobian entries that come out as the constants
0 and 1 become the structural-sparsity pattern the Riccati kernels exploit.

The emitted code performs the recorded operations in the recorded order with
no fused multiply-adds (the library is compiled with ``-ffp-contract=off``), i.e.
the same roundings as numpy up to the last bit of the transcendental functions.

Two other common idioms are supported by letting them through while the methods
are being recorded (``_recording_numpy``): arrays made with ``np.zeros`` /
``np.eye`` / ... and filled by item assignment, and ``np.asarray(x, dtype=float)``.

What cannot be traced (the caller then gets a ``TraceError`` that says so and
can write the snippets by hand, ``systems.DeviceSystem``): data-dependent
branches or comparisons, conversion of state values to Python floats/ints
(``float(x[0])``), numpy functions imported by name (``from numpy import
zeros``) in the idioms above, and functions outside the table below.
"""
import contextlib
import math
import threading

import numpy as np


class TraceError(TypeError):
    pass


_UNARY = {  # numpy ufunc method name -> device expression
    "exp": "exp({0})", "log": "log({0})", "sqrt": "sqrt({0})", "tanh": "tanh({0})",
    "sinh": "sinh({0})", "cosh": "cosh({0})", "arctan": "atan({0})", "arcsin": "asin({0})",
    "arccos": "acos({0})", "abs": "fabs_({0})", "neg": "(-{0})",
}
_BINARY = {"add": "({0} + {1})", "sub": "({0} - {1})", "mul": "({0} * {1})", "div": "({0} / {1})",
           "arctan2": "atan2({0}, {1})", "pow": "pow({0}, {1})"}
_NUMPY_FN = {"exp": np.exp, "log": np.log, "sqrt": np.sqrt, "tanh": np.tanh, "sinh": np.sinh, "cosh": np.cosh,
             "arctan": np.arctan, "arcsin": np.arcsin, "arccos": np.arccos, "abs": np.abs, "neg": np.negative,
             "sin": np.sin, "cos": np.cos, "tan": np.tan,
             "add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "arctan2": np.arctan2,
             "pow": np.power}


def _is_number(v):
    return isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool)


class Sym:
    """One recorded scalar: a leaf (``x[i]`` / ``u[j]``) or an operation on earlier ones."""

    __slots__ = ("op", "args", "graph", "index")

    def __init__(self, graph, op, args):
        self.graph, self.op, self.args = graph, op, args
        self.index = len(graph.nodes)
        graph.nodes.append(self)

    # -- construction with hash-consing and the exact identities x*1, x+0, x-0, x/1
    def _make(self, op, *args):
        g = self.graph
        key = (op,) + tuple(("c", float(a)) if _is_number(a) else ("s", a.index) for a in args)
        node = g.memo.get(key)
        if node is None:
            node = g.memo[key] = Sym(g, op, args)
        return node

    def _bin(self, op, a, b):
        for v in (a, b):
            if not (_is_number(v) or isinstance(v, Sym)):
                return NotImplemented
        if op == "mul":
            if _is_number(a) and float(a) == 1.0:
                return b
            if _is_number(b) and float(b) == 1.0:
                return a
        if op in ("add", "sub") and _is_number(b) and float(b) == 0.0:
            return a
        if op == "add" and _is_number(a) and float(a) == 0.0:
            return b
        if op == "div" and _is_number(b) and float(b) == 1.0:
            return a
        return self._make(op, a, b)

    def __add__(self, o): return self._bin("add", self, o)
    def __radd__(self, o): return self._bin("add", o, self)
    def __sub__(self, o): return self._bin("sub", self, o)
    def __rsub__(self, o): return self._bin("sub", o, self)
    def __mul__(self, o): return self._bin("mul", self, o)
    def __rmul__(self, o): return self._bin("mul", o, self)
    def __truediv__(self, o): return self._bin("div", self, o)
    def __rtruediv__(self, o): return self._bin("div", o, self)
    def __neg__(self): return self._make("neg", self)
    def __pos__(self): return self
    def __abs__(self): return self._make("abs", self)

    def __pow__(self, e):
        if _is_number(e) and float(e) == int(e) and 1 <= int(e) <= 4:
            out = self
            for _ in range(int(e) - 1):
                out = out * self
            return out
        return self._bin("pow", self, e)

    def __rpow__(self, b): return self._bin("pow", b, self)

    # numpy ufuncs applied to object arrays call the method of the same name
    def sin(self): return self._make("sin", self)
    def cos(self): return self._make("cos", self)
    def tan(self): return self._make("tan", self)
    def exp(self): return self._make("exp", self)
    def log(self): return self._make("log", self)
    def sqrt(self): return self._make("sqrt", self)
    def tanh(self): return self._make("tanh", self)
    def sinh(self): return self._make("sinh", self)
    def cosh(self): return self._make("cosh", self)
    def arctan(self): return self._make("arctan", self)
    def arcsin(self): return self._make("arcsin", self)
    def arccos(self): return self._make("arccos", self)
    def arctan2(self, o): return self._bin("arctan2", self, o)
    def square(self): return self * self
    def conjugate(self): return self

    def _refuse(self, *_):
        raise TraceError("the model compares or converts a state/control value (data-dependent branch, float(), "
                         "int()): it is not straight-line arithmetic and cannot be traced; "
                         "derive it from systems.DeviceSystem and write the device snippets by hand")

    # (== and != too: the default identity comparison would answer False / True as a plain bool and
    # silently record one branch of a guard such as ``if v == 0.0``)
    __lt__ = __le__ = __gt__ = __ge__ = __eq__ = __ne__ = __bool__ = __float__ = __int__ = __index__ = _refuse
    __hash__ = object.__hash__


class _Graph:
    def __init__(self, n, m):
        self.nodes, self.memo = [], {}
        self.x = np.empty(n, dtype=object)
        self.u = np.empty(m, dtype=object)
        for i in range(n):
            self.x[i] = Sym(self, "x", (i,))
        for j in range(m):
            self.u[j] = Sym(self, "u", (j,))

    def evaluate(self, outputs, x, u):
        """Numeric value of the recorded outputs at one point (host check of the record)."""
        val = {}

        def get(a):
            return float(a) if _is_number(a) else val[a.index]

        for node in self.nodes:
            if node.op == "x":
                val[node.index] = float(x[node.args[0]])
            elif node.op == "u":
                val[node.index] = float(u[node.args[0]])
            else:
                val[node.index] = float(_NUMPY_FN[node.op](*[get(a) for a in node.args]))
        return np.array([get(o) for o in outputs], dtype=float)


def _literal(v):
    v = float(v)
    if not math.isfinite(v):
        raise TraceError("the model produces a non-finite constant")
    return f"T({v!r})"


def _emit(graph, assignments):
    """Device statements computing the given ``(target, value)`` assignments."""
    needed, stack = set(), [v for _, v in assignments if isinstance(v, Sym)]
    while stack:
        node = stack.pop()
        if node.index in needed:
            continue
        needed.add(node.index)
        stack.extend(a for a in node.args if isinstance(a, Sym))
    name = {}
    lines = []

    def ref(a):
        return _literal(a) if _is_number(a) else name[a.index]

    trig_done = {}
    for node in graph.nodes:           # creation order is a topological order
        if node.index not in needed:
            continue
        if node.op == "x":
            name[node.index] = f"x[{node.args[0]}]"
        elif node.op == "u":
            name[node.index] = f"u[{node.args[0]}]"
        elif node.op in ("sin", "cos", "tan"):
            arg = node.args[0]
            if arg.index not in trig_done:   # one sincos per distinct angle
                k = arg.index
                lines.append(f"T s{k}, c{k}; sincos_<FAST>({ref(arg)}, s{k}, c{k}, bad);")
                trig_done[arg.index] = k
            k = trig_done[arg.index]
            if node.op == "tan":
                lines.append(f"const T v{node.index} = s{k} / c{k};")
                name[node.index] = f"v{node.index}"
            else:
                name[node.index] = f"{'s' if node.op == 'sin' else 'c'}{k}"
        else:
            fmt = _UNARY.get(node.op) or _BINARY[node.op]
            expr = fmt.format(*[ref(a) for a in node.args])
            lines.append(f"const T v{node.index} = {expr};")
            name[node.index] = f"v{node.index}"
    for target, value in assignments:
        lines.append(f"{target} = {ref(value)};")
    return "\n".join(lines)


def _flat(out, count, what):
    a = np.asarray(out, dtype=object).reshape(-1)
    if a.size != count:
        raise TraceError(f"{what} returned {a.size} values, expected {count}")
    for v in a:
        if not (_is_number(v) or isinstance(v, Sym)):
            raise TraceError(f"{what} returned a value of type {type(v).__name__}")
    return list(a)


_FLOAT_DTYPES = (float, np.float64, np.float32, "float", "float64", "float32", "d", "f8")
_ALLOCATORS = ("zeros", "ones", "empty", "full", "eye", "identity", "zeros_like", "ones_like", "empty_like", "full_like")
_CONVERTERS = ("asarray", "array", "ascontiguousarray", "asanyarray")
_patch_lock = threading.Lock()


def _holds_sym(obj):
    if isinstance(obj, Sym):
        return True
    if isinstance(obj, np.ndarray):
        return obj.dtype == object and any(isinstance(v, Sym) for v in obj.reshape(-1))
    if isinstance(obj, (list, tuple)):
        return any(_holds_sym(v) for v in obj)
    return False


@contextlib.contextmanager
def _recording_numpy():
    """While the methods are being recorded, the two idioms that would otherwise stop a recording
    scalar at the door are let through: arrays allocated with ``np.zeros`` / ``np.eye`` / ... and
    filled by item assignment hold objects instead of floats, and ``np.asarray(x, dtype=float)``
    (and friends) of recording scalars keeps them instead of converting.  Only the attributes of
    the ``numpy`` module that user code reaches as ``np.<name>`` are replaced, only for the
    duration of the three calls, under a lock, and they behave differently only when called from
    the recording thread (other threads get numpy's own behaviour); numpy's own internals do not go
    through them.  The record is validated against the unmodified methods afterwards."""
    saved = {}
    owner = threading.get_ident()

    def allocator(fn):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            if threading.get_ident() != owner:
                return out
            return out.astype(object) if isinstance(out, np.ndarray) and out.dtype.kind in "fiu" else out
        return wrapped

    def converter(fn):
        def wrapped(*a, **k):
            # the converted object is the first positional argument or numpy's keyword for it
            obj = a[0] if a else k.get("object", k.get("a"))
            dt = k.get("dtype", a[1] if len(a) > 1 else None)
            if threading.get_ident() == owner and _holds_sym(obj) and (dt is None or dt in _FLOAT_DTYPES or dt is object):
                k.pop("dtype", None)
                return fn(*a[:1], *a[2:], dtype=object, **k)
            return fn(*a, **k)
        return wrapped

    with _patch_lock:
        try:
            for name in _ALLOCATORS:
                saved[name] = getattr(np, name)
                setattr(np, name, allocator(saved[name]))
            for name in _CONVERTERS:
                saved[name] = getattr(np, name)
                setattr(np, name, converter(saved[name]))
            yield
        finally:
            for name, fn in saved.items():
                setattr(np, name, fn)


def trace_system(system, check_points=8, seed=0):
    """Record ``model_f``, ``compute_df_dx`` and ``compute_df_du`` of ``system``.

    Returns ``dict(step=..., jac=..., fx_pattern=..., fu_pattern=...)``: the device snippets and
    sparsity patterns for ``jit.model_source``.  Raises ``TraceError`` if the methods are not
    straight-line arithmetic, or if the record disagrees with the methods at random points."""
    n, m = int(system.state_size), int(system.control_size)
    g = _Graph(n, m)
    try:
        with _recording_numpy():
            f = _flat(system.model_f(g.x.copy(), g.u.copy()), n, "model_f")
            fx = _flat(system.compute_df_dx(g.x.copy(), g.u.copy()), n * n, "compute_df_dx")
            fu = _flat(system.compute_df_du(g.x.copy(), g.u.copy()), n * m, "compute_df_du")
    except TraceError:
        raise
    except Exception as e:  # numpy refuses an operation on the recording objects
        raise TraceError(f"{type(system).__name__}: the numpy methods cannot be traced ({type(e).__name__}: {e}); "
                         "derive the model from systems.DeviceSystem and write the device snippets by hand") from e
    rng = np.random.default_rng(seed)
    for _ in range(check_points):
        xs, us = rng.normal(size=n), rng.normal(size=m)
        for rec, meth, what in ((f, system.model_f, "model_f"), (fx, system.compute_df_dx, "compute_df_dx"),
                                (fu, system.compute_df_du, "compute_df_du")):
            want = np.asarray(meth(xs.copy(), us.copy()), dtype=float).reshape(-1)
            got = g.evaluate(rec, xs, us)
            if not np.allclose(got, want, rtol=1e-12, atol=1e-12, equal_nan=True):
                raise TraceError(f"{type(system).__name__}.{what} does not compute the same values when called again "
                                 "(hidden state or data-dependent behaviour); it cannot be traced")
    fxp = np.ones((n, n), dtype=np.int64)
    fup = np.ones((n, m), dtype=np.int64)
    jac = []
    for i in range(n):
        for j in range(n):
            v = fx[i * n + j]
            if _is_number(v) and float(v) == 0.0:
                fxp[i, j] = 0
            elif _is_number(v) and float(v) == 1.0:
                fxp[i, j] = 2
            else:
                jac.append((f"fx[{i}][{j}]", v))
        for j in range(m):
            v = fu[i * m + j]
            if _is_number(v) and float(v) == 0.0:
                fup[i, j] = 0
            else:
                jac.append((f"fu[{i}][{j}]", v))
    step = _emit(g, [(f"xn[{i}]", f[i]) for i in range(n)])
    return dict(step=step, jac=_emit(g, jac), fx_pattern=fxp, fu_pattern=fup)


def traced_model(system, dtype=np.float64, diag=True, has_barrier=False):
    """Trace, compile (cached) and register the dynamics of ``system``; returns the model id.
    ``has_barrier``: the state starts with (px, py, v), the layout ``constraints.py:12`` is written
    for (needed by the barrier mode and by the sequential-QP solver)."""
    from . import jit
    t = trace_system(system)
    variant = ("f64" if np.dtype(dtype) == np.float64 else "f32", bool(diag))
    src = jit.model_source(system.state_size, system.control_size, t["step"], t["jac"], t["fx_pattern"],
                           t["fu_pattern"], has_barrier=bool(has_barrier), variants=(variant,))
    return jit.load_model(jit.build_model(src))
