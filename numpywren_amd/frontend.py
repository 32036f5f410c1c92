"""LambdaPACK front end: Python-embedded DSL  ->  loop-nest IR.

Counterpart of the reference's frontend.py (LambdaPackParse / LambdaPackTypeCheck /
BackendGenerate, reference numpywren/frontend.py:223-843), redesigned for a static executor:
instead of a symbolic (sympy) representation that is solved per task at run time, the DSL is
lowered to a small tree of `For` / `If` / `Let` / `Call` nodes whose index expressions are
compiled to Python code objects.  compiler.py then expands the tree by concrete interpretation.

Accepted subset (the same one the reference documents in its README "lambdapack restrictions"):
  for v in range(a[, b[, c]]):           loops with affine or non-affine integer bounds
  if <static condition>: ... else: ...   conditions over loop variables / integer arguments
  name = <scalar expression>             scalar bindings (N_tree = ceiling(log(N - i)/log(2)))
  M[i, j], N[k] = kernel(A[i, k], 2.0, B[k, j], kw=...)   one kernel call per statement (SSA tiles)
Scalar expressions: + - * / // % **, comparisons, and/or/not, ceiling floor log log2 min max abs int.
Statements are numbered in order of appearance (`expr_idx`), exactly like the reference
(frontend.py:779-784), which is what task ids (expr_idx, {loop var: value}) refer to.
"""
import ast
import inspect
import math
import operator as _operator
import textwrap

from .exceptions import LambdaPackParsingException, LambdaPackTypeException


# ------------------------------------------------------------------------------------------------
# exact integer logarithm ratios: ceiling(log(8)/log(2)) must be 3, not ceil(3.0000000000000004)
# ------------------------------------------------------------------------------------------------
class _Log(object):
    __slots__ = ("x",)

    def __init__(self, x):
        self.x = x

    def __float__(self):
        return math.log(self.x)

    def _ratio(self, num, den):
        r = math.log(num) / math.log(den)
        k = int(round(r))
        if isinstance(num, int) and isinstance(den, int) and k >= 0 and den ** k == num:
            return k
        return r

    def __truediv__(self, other):
        if isinstance(other, _Log):
            return self._ratio(self.x, other.x)
        return float(self) / other

    def __rtruediv__(self, other):
        return other / float(self)

    def __mul__(self, other):
        return float(self) * float(other)

    __rmul__ = __mul__

    def __add__(self, other):
        return float(self) + float(other)

    __radd__ = __add__

    def __sub__(self, other):
        return float(self) - float(other)

    def __rsub__(self, other):
        return float(other) - float(self)


def _log(x):
    if x <= 0:
        raise LambdaPackTypeException(f"log of non-positive value {x} in a LambdaPACK index expression")
    return _Log(x)


def _ceiling(x):
    return int(math.ceil(float(x) - 1e-12)) if not isinstance(x, int) else x


def _floor(x):
    return int(math.floor(float(x) + 1e-12)) if not isinstance(x, int) else x


SCALAR_FUNCTIONS = {
    "ceiling": _ceiling, "ceil": _ceiling, "floor": _floor, "log": _log,
    "log2": lambda x: _Log(x) / _Log(2), "min": min, "max": max, "abs": abs, "int": int, "float": float,
    "True": True, "False": False,
}


# ------------------------------------------------------------------------------------------------
# IR
# ------------------------------------------------------------------------------------------------
class Expr(object):
    """A scalar expression compiled to a code object; evaluated against the loop environment."""
    __slots__ = ("src", "code", "const")

    def __init__(self, node):
        self.src = ast.unparse(node) if hasattr(ast, "unparse") else ""
        self.code = compile(ast.fix_missing_locations(ast.Expression(body=node)), "<lambdapack>", "eval")
        self.const = None
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float, bool)):
            self.const = node.value

    def eval(self, env):
        if self.const is not None:
            return self.const
        return eval(self.code, SCALAR_FUNCTIONS, env)

    def __repr__(self):
        return f"Expr({self.src})"


class IndexExpr(object):
    """M[e0, e1, ...]"""
    __slots__ = ("matrix_name", "indices")

    def __init__(self, matrix_name, indices):
        self.matrix_name = matrix_name
        self.indices = indices

    def __repr__(self):
        return f"{self.matrix_name}[{', '.join(i.src for i in self.indices)}]"


class Call(object):
    __slots__ = ("expr_idx", "kernel_name", "outputs", "args", "kwargs", "loop_vars", "lineno")

    def __init__(self, expr_idx, kernel_name, outputs, args, kwargs, loop_vars, lineno):
        self.expr_idx = expr_idx
        self.kernel_name = kernel_name
        self.outputs = outputs      # [IndexExpr]
        self.args = args            # [IndexExpr | Expr]
        self.kwargs = kwargs        # {name: Expr}
        self.loop_vars = loop_vars  # names of the enclosing loop variables, outermost first
        self.lineno = lineno

    def reads(self):
        return [a for a in self.args if isinstance(a, IndexExpr)]


class For(object):
    __slots__ = ("var", "start", "stop", "step", "body", "calls")

    def __init__(self, var, start, stop, step, body):
        self.var, self.start, self.stop, self.step, self.body = var, start, stop, step, body
        self.calls = frozenset()


class If(object):
    __slots__ = ("test", "body", "orelse", "calls")

    def __init__(self, test, body, orelse):
        self.test, self.body, self.orelse = test, body, orelse
        self.calls = frozenset()


class Let(object):
    __slots__ = ("name", "value")

    def __init__(self, name, value):
        self.name, self.value = name, value


class ProgramIR(object):
    """Parsed DSL function: argument names, statement tree, the flat list of kernel calls."""

    def __init__(self, name, arg_names, body, calls, globals_):
        self.name = name
        self.arg_names = arg_names
        self.body = body
        self.calls = calls          # [Call] indexed by expr_idx
        self.globals = globals_


# ------------------------------------------------------------------------------------------------
# parser
# ------------------------------------------------------------------------------------------------
class _Parser(object):
    def __init__(self):
        self.calls = []

    def fail(self, node, msg):
        raise LambdaPackParsingException(f"line {getattr(node, 'lineno', '?')}: {msg}")

    def index_expr(self, node):
        if not (isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name)):
            self.fail(node, "expected a matrix index expression like M[i, j]")
        sl = node.slice
        if isinstance(sl, ast.Index):  # python < 3.9
            sl = sl.value
        elts = sl.elts if isinstance(sl, ast.Tuple) else [sl]
        for e in elts:
            if isinstance(e, ast.Slice):
                self.fail(node, "slices are not allowed in tile indices")
        return IndexExpr(node.value.id, [Expr(e) for e in elts])

    def block(self, stmts, loop_vars):
        out = []
        for st in stmts:
            if isinstance(st, ast.Expr) and isinstance(st.value, ast.Constant):
                continue  # docstring / bare literal
            if isinstance(st, ast.Pass):
                continue
            if isinstance(st, ast.For):
                out.append(self.for_(st, loop_vars))
            elif isinstance(st, ast.If):
                node = If(Expr(st.test), self.block(st.body, loop_vars), self.block(st.orelse, loop_vars))
                out.append(node)
            elif isinstance(st, ast.Assign):
                out.append(self.assign(st, loop_vars))
            elif isinstance(st, ast.AnnAssign) and st.value is not None and isinstance(st.target, ast.Name):
                out.append(Let(st.target.id, Expr(st.value)))
            else:
                self.fail(st, f"unsupported statement {type(st).__name__}")
        return out

    def for_(self, st, loop_vars):
        if st.orelse:
            self.fail(st, "for/else is not supported")
        if not isinstance(st.target, ast.Name):
            self.fail(st, "loop target must be a plain name")
        it = st.iter
        if not (isinstance(it, ast.Call) and isinstance(it.func, ast.Name) and it.func.id == "range"
                and 1 <= len(it.args) <= 3 and not it.keywords):
            self.fail(st, "loops must iterate over range(...)")
        a = it.args
        zero, one = ast.Constant(value=0), ast.Constant(value=1)
        if len(a) == 1:
            start, stop, step = zero, a[0], one
        elif len(a) == 2:
            start, stop, step = a[0], a[1], one
        else:
            start, stop, step = a
        body = self.block(st.body, loop_vars + [st.target.id])
        return For(st.target.id, Expr(start), Expr(stop), Expr(step), body)

    def assign(self, st, loop_vars):
        if len(st.targets) != 1:
            self.fail(st, "chained assignment is not supported")
        tgt = st.targets[0]
        if isinstance(tgt, ast.Name):
            return Let(tgt.id, Expr(st.value))
        outs = tgt.elts if isinstance(tgt, ast.Tuple) else [tgt]
        outputs = [self.index_expr(o) for o in outs]
        call = st.value
        if not (isinstance(call, ast.Call) and isinstance(call.func, ast.Name)):
            self.fail(st, "the right-hand side of a tile assignment must be a kernel call")
        args = []
        for a in call.args:
            if isinstance(a, ast.Subscript):
                args.append(self.index_expr(a))
            elif isinstance(a, ast.Starred):
                self.fail(st, "starred arguments are not supported")
            else:
                args.append(Expr(a))
        kwargs = {}
        for kw in call.keywords:
            if kw.arg is None:
                self.fail(st, "**kwargs are not supported")
            kwargs[kw.arg] = Expr(kw.value)
        c = Call(len(self.calls), call.func.id, outputs, args, kwargs, list(loop_vars), st.lineno)
        self.calls.append(c)
        return c


def _annotate(nodes):
    """Fill For.calls / If.calls with the expr_idx set of their subtrees (lets the expander skip loops
    that contain no statement of interest)."""
    acc = set()
    for n in nodes:
        if isinstance(n, Call):
            acc.add(n.expr_idx)
        elif isinstance(n, For):
            n.calls = frozenset(_annotate(n.body))
            acc |= n.calls
        elif isinstance(n, If):
            n.calls = frozenset(_annotate(n.body) | _annotate(n.orelse))
            acc |= n.calls
    return acc


def parse(function):
    """DSL function object (or its source string) -> ProgramIR."""
    if isinstance(function, str):
        src, globs = function, {}
    else:
        src, globs = inspect.getsource(function), getattr(function, "__globals__", {})
    tree = ast.parse(textwrap.dedent(src))
    fdefs = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
    if len(fdefs) != 1:
        raise LambdaPackParsingException("expected exactly one function definition")
    f = fdefs[0]
    if f.args.vararg or f.args.kwarg or f.args.kwonlyargs:
        raise LambdaPackParsingException("LambdaPACK programs take positional arguments only")
    p = _Parser()
    body = p.block(f.body, [])
    _annotate(body)
    return ProgramIR(f.name, [a.arg for a in f.args.args], body, p.calls, globs)


def resolve_kernel(name, globals_=None, extra=None):
    """Kernel lookup with the reference's precedence (frontend.py:12-15: `from numpywren.kernels
    import *` followed by `from operator import *`, so `mul` / `add` are the operator-module
    functions), then the DSL function's own globals."""
    if extra and name in extra:
        return extra[name]
    if hasattr(_operator, name) and not name.startswith("_"):
        return getattr(_operator, name)
    from . import kernels
    if hasattr(kernels, name):
        return getattr(kernels, name)
    if globals_ and name in globals_:
        return globals_[name]
    raise LambdaPackParsingException(f"unknown kernel '{name}'")
