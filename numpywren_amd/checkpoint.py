"""Checkpoint / resume of a LambdaPACK run on disk.

In the reference a run needs no checkpoint: every tile is an S3 object, every node / edge state a Redis key, every task
idempotent -- a program interrupted anywhere is resumed by starting workers again ("Checkpoint / resume:
implicit: all state is durable").  Here the tiles live in HBM and the run state in one process, so the durable form is explicit:

    checkpoint.save(program, root)      tiles of the program's matrices as the reference's objects (tile_io.export_matrix:
                                        `np.save` bodies + header JSON under <root>/<bucket>/<key>/...), and the finished nodes
                                        as <root>/<bucket>/lambdapack/<name>/state.json -- next to where the reference keeps a
                                        run's records (s3://bucket/lambdapack/<hash>/, job_runner.py:359-364)
    checkpoint.load(program, root)      in a new process: the same program built again (same matrices, same sizes) gets its tiles
                                        back and its dependency accounting replayed for the finished nodes
                                        (LambdaPackProgram.resume); job_runner.lambdapack_run then continues where the run stopped.

`save` is atomic per object and as a whole: every tile body and the state record are written to a temporary name and renamed
into place, the state record last, and it lists the tile objects that belong to this checkpoint -- `load` restores exactly
those, so a crash in the middle of a save leaves the previous checkpoint readable, and tiles of intermediates that were
reclaimed since an earlier save into the same root are not brought back.  It refuses a program in EXCEPTION state (its tiles
are not a consistent state of the run) and a distributed run (each rank only holds its own tiles; SURVEY 8e keeps the
multi-GPU path free of a durable form).  With `executor.fuse_gemm_reduction` the partial sums of unfinished C tiles live in
accumulators outside the tile store: the products that went into a still-open accumulator are recorded as NOT finished, so a
resumed run forms those sums again from the start (the accumulators themselves are dropped by `resume`).

A run is checkpointed between tasks: after lambdapack_run returned on its time limit (the program stays RUNNING), or after
program.wait() on a finished run (then resume has nothing left to do).  Tiles come back as plain tiles: what the backend knew
beyond their bytes (an R factor's `upper` flag, a factor's cached block inverses) is recomputed or does without.
"""
import json
import os

from . import lambdapack as lp
from . import tile_io


def _state_path(root, program, name):
    bucket = next(iter(program.program.matrices.values())).bucket if program.program.matrices else "hbm"
    return os.path.join(root, bucket, "lambdapack", name, "state.json")


def finished_nodes(program):
    """[(expr_idx, var_values)] of the tasks whose status is FINISHED, in program order."""
    return [(t.expr_idx, dict(t.vars)) for t in program.program.tasks
            if program.get_node_status(t.expr_idx, t.vars) == lp.NS.FINISHED]


def _write_atomic(path, data, mode):
    tmp = path + ".tmp"
    with open(tmp, mode) as f:
        f.write(data)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def _open_fusion_members(program):
    """Task indices whose result only exists inside a still-open accumulator of job_runner.ReductionFusion."""
    acc = program.__dict__.get("_fusion_acc")
    if not acc:
        return set()
    from .job_runner import ReductionFusion
    fusion = ReductionFusion(program.program)
    pending = set(acc.keys())
    return {idx for idx, root in fusion.root_of.items() if root in pending}


def save(program, root, name="checkpoint", include_inputs=True):
    """Write the run's durable state under `root`.  Returns {"tiles": n, "finished": k, "tasks": total}."""
    finish = getattr(program, "_finish", None)
    if finish is not None:          # a wait=False run still in flight: settle it first (the tiles must be final)
        finish()
    from .device import get_backend
    try:
        get_backend().synchronize()
    except Exception:
        pass
    if program.program_status() == lp.PS.EXCEPTION:
        raise ValueError("checkpoint.save: the program is in EXCEPTION state; its tiles are not a resumable state of the run")
    if int(getattr(program, "_distributed_world", 1) or 1) > 1:
        raise NotImplementedError("checkpoint.save: a distributed run is not checkpointed (each rank holds only its own tiles)")
    compiled = program.program
    listed = {}
    tiles = 0
    for mname, m in compiled.matrices.items():
        if not include_inputs and mname in compiled.inputs:
            continue
        keys = tile_io.export_matrix(m, root, atomic=True, return_keys=True)
        listed[mname] = keys
        tiles += len(keys)
    open_members = _open_fusion_members(program)
    done = [(t.expr_idx, dict(t.vars)) for t in compiled.tasks
            if t.index not in open_members and program.get_node_status(t.expr_idx, t.vars) == lp.NS.FINISHED]
    state = {"tasks": len(compiled.tasks), "finished": [[int(e), sorted((str(k), int(x)) for k, x in v.items())] for e, v in done],
             "matrices": {n: {"key": m.key, "shape": [int(s) for s in m.shape], "shard_sizes": [int(s) for s in m.shard_sizes]}
                          for n, m in compiled.matrices.items()},
             "tiles": listed,
             "status": program.program_status().name}
    path = _state_path(root, program, name)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    _write_atomic(path, json.dumps(state), "w")        # last: a state record always describes complete tile objects
    return {"tiles": tiles, "finished": len(done), "tasks": len(compiled.tasks)}


def load(program, root, name="checkpoint"):
    """Bring a saved run back into `program` (built again over matrices of the same keys, shapes and shard sizes): tiles into
    the store, finished nodes into the dependency accounting.  Returns the number of tasks still to run."""
    path = _state_path(root, program, name)
    with open(path) as f:
        state = json.load(f)
    compiled = program.program
    if state["tasks"] != len(compiled.tasks):
        raise ValueError("checkpoint of a program with {0} tasks, this one has {1}".format(state["tasks"], len(compiled.tasks)))
    for n, meta in state["matrices"].items():
        m = compiled.matrices.get(n)
        if m is None or m.key != meta["key"] or [int(s) for s in m.shape] != meta["shape"] or \
                [int(s) for s in m.shard_sizes] != meta["shard_sizes"]:
            raise ValueError("checkpoint matrix {0!r} ({1}) does not match this program's".format(n, meta))
    listed = state.get("tiles")
    for n, m in compiled.matrices.items():
        if listed is None:
            tile_io.restore(m, root)                       # (a checkpoint written before the tile list existed)
        elif n in listed:
            tile_io.restore(m, root, only=set(listed[n]))
    program.resume([(e, dict(v)) for e, v in state["finished"]])
    return len(compiled.tasks) - len(state["finished"])
