"""LambdaPACK compiler: loop-nest IR  ->  static task DAG.

Counterpart of the reference's compiler.py (reference numpywren/compiler.py:25-800).  The
reference keeps the DAG *implicit* and answers find_children / find_parents by solving index
equations with sympy on every post_op (~17 ms per node, ~28 ms per task; SURVEY.md section 2
row 9) -- fine for S3/Lambda latencies, three orders of magnitude too slow next to a GPU
kernel.  Here the program is *expanded once* by concrete interpretation of the loop nest:
every task instance records the tiles it reads and writes, edges follow from "who wrote the
tile I read" (LambdaPACK programs are SSA: one writer per tile), and the same
find_children / find_parents / starters / num_terminators / eval_expr interface is served from
hash tables.  Starters and terminator counts only walk the loops that contain the relevant
statements, so very large programs (e.g. the 313-block Cholesky of the reference's tests,
5 M tasks) can still be counted without materialising the graph.

Public surface (same names as the reference): lpcompile, lpcompile_for_execution,
CompiledLambdaPackProgram{starters, num_terminators, inputs, outputs, find_children,
find_parents, is_terminator, eval_expr}, walk_program, find_starters, find_terminators,
find_children, find_parents.
"""
from . import frontend
from . import lambdapack as lp
from .exceptions import LambdaPackBackendGenerationException
from .frontend import Call, For, If, IndexExpr, Let


def _to_int(v, what):
    if isinstance(v, bool):
        return int(v)
    if isinstance(v, int):
        return v
    f = float(v)
    if abs(f - round(f)) > 1e-9:
        raise LambdaPackBackendGenerationException(f"{what} evaluated to the non-integer value {v}")
    return int(round(f))


def node_key(expr_idx, var_values):
    return (int(expr_idx), tuple(sorted((str(k), int(v)) for k, v in var_values.items())))


class Task(object):
    """One node of the DAG: statement `expr_idx` instantiated at loop values `vars`."""
    __slots__ = ("expr_idx", "vars", "key", "call", "reads", "consts", "arg_kinds", "writes", "kwargs",
                 "parents", "children", "index")

    def __init__(self, call, var_values):
        self.call = call
        self.expr_idx = call.expr_idx
        self.vars = var_values
        self.key = node_key(call.expr_idx, var_values)
        self.reads = []       # [(matrix_name, idx tuple)] in argument order
        self.consts = []      # scalar arguments
        self.arg_kinds = []   # per positional argument: ("tile", i) into reads | ("const", i) into consts
        self.writes = []      # [(matrix_name, idx tuple)] in output order
        self.kwargs = {}
        self.parents = []
        self.children = []
        self.index = -1

    @property
    def node(self):
        return (self.expr_idx, dict(self.vars))

    def __repr__(self):
        return f"Task({self.expr_idx}, {self.vars})"


class CompiledLambdaPackProgram(object):
    """A LambdaPACK program bound to concrete matrices and sizes."""

    def __init__(self, ir, args, inputs, outputs, kernels=None):
        if len(args) != len(ir.arg_names):
            raise LambdaPackBackendGenerationException(
                f"{ir.name} expects {len(ir.arg_names)} arguments ({', '.join(ir.arg_names)}), got {len(args)}")
        self.ir = ir
        self.inputs = list(inputs) if inputs is not None else []
        self.outputs = list(outputs) if outputs is not None else []
        self.bindings = dict(zip(ir.arg_names, args))
        self.matrices = {n: a for n, a in self.bindings.items() if hasattr(a, "get_block")}
        self.scalars = {n: a for n, a in self.bindings.items() if n not in self.matrices}
        self.remote_calls = {c.expr_idx: c for c in ir.calls}
        self._kernels = {c.expr_idx: frontend.resolve_kernel(c.kernel_name, ir.globals, kernels) for c in ir.calls}
        for c in ir.calls:
            for ie in c.outputs + c.reads():
                if ie.matrix_name not in self.matrices:
                    raise LambdaPackBackendGenerationException(
                        f"line {c.lineno}: '{ie.matrix_name}' is indexed like a matrix but is not a BigMatrix argument")
        self._tasks = None
        self._by_key = None
        self._starters = None
        self._num_terminators = None

    # ------------------------------------------------------------------ expansion
    def _walk(self, nodes, env, wanted, emit):
        for n in nodes:
            if isinstance(n, Call):
                if wanted is None or n.expr_idx in wanted:
                    emit(n, env)
            elif isinstance(n, For):
                if wanted is not None and not (n.calls & wanted):
                    continue
                start = _to_int(n.start.eval(env), "range start")
                stop = _to_int(n.stop.eval(env), "range stop")
                step = _to_int(n.step.eval(env), "range step")
                had = n.var in env
                old = env.get(n.var)
                for v in range(start, stop, step):
                    env[n.var] = v
                    self._walk(n.body, env, wanted, emit)
                if had:
                    env[n.var] = old
                else:
                    env.pop(n.var, None)
            elif isinstance(n, If):
                if wanted is not None and not (n.calls & wanted):
                    continue
                self._walk(n.body if n.test.eval(env) else n.orelse, env, wanted, emit)
            elif isinstance(n, Let):
                env[n.name] = n.value.eval(env)

    def _instantiate(self, call, env):
        t = Task(call, {v: int(env[v]) for v in call.loop_vars})
        for a in call.args:
            if isinstance(a, IndexExpr):
                idx = tuple(_to_int(i.eval(env), f"index of {a.matrix_name}") for i in a.indices)
                t.arg_kinds.append(("tile", len(t.reads)))
                t.reads.append((a.matrix_name, idx))
            else:
                t.arg_kinds.append(("const", len(t.consts)))
                t.consts.append(a.eval(env))
        for o in call.outputs:
            t.writes.append((o.matrix_name, tuple(_to_int(i.eval(env), f"index of {o.matrix_name}") for i in o.indices)))
        t.kwargs = {k: v.eval(env) for k, v in call.kwargs.items()}
        return t

    def _expand(self):
        if self._tasks is not None:
            return
        tasks = []
        self._walk(self.ir.body, dict(self.scalars), None, lambda c, env: tasks.append(self._instantiate(c, env)))
        by_key = {}
        writer = {}
        for i, t in enumerate(tasks):
            t.index = i
            if t.key in by_key:
                raise LambdaPackBackendGenerationException(f"statement {t.expr_idx} instantiated twice at {t.vars}")
            by_key[t.key] = t
            for w in t.writes:
                if w in writer:
                    raise LambdaPackBackendGenerationException(
                        f"tile {w[0]}{list(w[1])} is written by both {writer[w]} and {t}: LambdaPACK programs "
                        "must be single-assignment")
                writer[w] = t
        for t in tasks:
            seen = set()
            for r in t.reads:
                p = writer.get(r)
                if p is not None and p.index not in seen and p is not t:
                    seen.add(p.index)
                    t.parents.append(p)
                    p.children.append(t)
        self._tasks, self._by_key, self._writer = tasks, by_key, writer

    @property
    def tasks(self):
        self._expand()
        return self._tasks

    def task(self, expr_idx, var_values):
        self._expand()
        try:
            return self._by_key[node_key(expr_idx, var_values)]
        except KeyError:
            raise KeyError(f"({expr_idx}, {var_values}) is not a node of program {self.ir.name}")

    def writer_of(self, matrix_name, idx):
        self._expand()
        return self._writer.get((matrix_name, tuple(idx)))

    # ------------------------------------------------------------------ reference interface
    def _reads_only_from(self, call, names):
        return all(ie.matrix_name in names for ie in call.reads())

    def _writes_to(self, call, names):
        return any(o.matrix_name in names for o in call.outputs)

    @property
    def starters(self):
        """Every instance of every statement that reads only from the input matrices (reference
        compiler.py:709-719) -- including, as in the reference, instances that do have parents when a
        program reads and writes the same input matrix."""
        if self._starters is None:
            wanted = frozenset(c.expr_idx for c in self.ir.calls if self._reads_only_from(c, set(self.inputs)))
            out = []
            if wanted:
                self._walk(self.ir.body, dict(self.scalars), wanted,
                           lambda c, env: out.append((c.expr_idx, {v: int(env[v]) for v in c.loop_vars})))
            self._starters = out
        return self._starters

    @property
    def num_terminators(self):
        """Number of instances of statements that write to an output matrix (reference compiler.py:721-732)."""
        if self._num_terminators is None:
            wanted = frozenset(c.expr_idx for c in self.ir.calls if self._writes_to(c, set(self.outputs)))
            count = [0]

            def bump(c, env):
                count[0] += 1

            if wanted:
                self._walk(self.ir.body, dict(self.scalars), wanted, bump)
            self._num_terminators = count[0]
        return self._num_terminators

    def find_children(self, i, value_map):
        return [c.node for c in self.task(i, value_map).children]

    def find_parents(self, i, value_map):
        return [p.node for p in self.task(i, value_map).parents]

    def is_terminator(self, i):
        return self._writes_to(self.remote_calls[i], set(self.outputs))

    def kernel(self, i):
        return self._kernels[i]

    def eval_expr(self, i, value_map):
        """The instruction block of one task: reads in argument order, one call, writes in output order
        (reference compiler.py:146-180)."""
        t = self.task(i, value_map)
        reads = [lp.RemoteRead(0, self.matrices[m], *idx) for m, idx in t.reads]
        argv = [reads[j] if kind == "tile" else t.consts[j] for kind, j in t.arg_kinds]
        symbols = [str(n) for n in range(len(argv))]
        call = lp.RemoteCall(0, self._kernels[i], argv, len(t.writes), symbols, **t.kwargs)
        writes = [lp.RemoteWrite(n + len(argv), self.matrices[m], call.results, n, *idx)
                  for n, (m, idx) in enumerate(t.writes)]
        return lp.InstructionBlock(reads + [call] + writes)

    def walk(self):
        return [t.node for t in self.tasks]


def lpcompile(function, kernels=None):
    """DSL function -> callable(*program_args) -> {expr_idx: statement}-like compiled program."""
    ir = frontend.parse(function)

    def f(*args, **kwargs):
        if kwargs:
            args = tuple(args) + tuple(kwargs[n] for n in ir.arg_names[len(args):])
        return CompiledLambdaPackProgram(ir, args, [], [], kernels=kernels)

    return f


def lpcompile_for_execution(function, inputs, outputs, kernels=None):
    ir = frontend.parse(function)

    def f(*args, **kwargs):
        if kwargs:
            args = tuple(args) + tuple(kwargs[n] for n in ir.arg_names[len(args):])
        return CompiledLambdaPackProgram(ir, args, inputs, outputs, kernels=kernels)

    return f


# ---- module-level helpers with the reference's call signatures (tests use these) ---------------
def walk_program(program):
    return program.walk()


def find_children(program, i, value_map):
    return program.find_children(i, value_map)


def find_parents(program, i, value_map):
    return program.find_parents(i, value_map)


def find_starters(program, input_matrices):
    p = CompiledLambdaPackProgram(program.ir, [program.bindings[n] for n in program.ir.arg_names],
                                  input_matrices, program.outputs)
    return p.starters


def find_terminators(program, output_matrices):
    p = CompiledLambdaPackProgram(program.ir, [program.bindings[n] for n in program.ir.arg_names],
                                  program.inputs, output_matrices)
    wanted = frozenset(c.expr_idx for c in p.ir.calls if p._writes_to(c, set(output_matrices)))
    out = []
    if wanted:
        p._walk(p.ir.body, dict(p.scalars), wanted,
                lambda c, env: out.append((c.expr_idx, {v: int(env[v]) for v in c.loop_vars})))
    return out
