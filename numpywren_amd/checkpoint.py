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
    compiled = program.program
    tiles = 0
    for mname, m in compiled.matrices.items():
        if not include_inputs and mname in compiled.inputs:
            continue
        tiles += tile_io.export_matrix(m, root)
    done = finished_nodes(program)
    state = {"tasks": len(compiled.tasks), "finished": [[int(e), sorted((str(k), int(x)) for k, x in v.items())] for e, v in done],
             "matrices": {n: {"key": m.key, "shape": [int(s) for s in m.shape], "shard_sizes": [int(s) for s in m.shard_sizes]}
                          for n, m in compiled.matrices.items()},
             "status": program.program_status().name}
    path = _state_path(root, program, name)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(state, f)
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
    for m in compiled.matrices.values():
        tile_io.restore(m, root)
    program.resume([(e, dict(v)) for e, v in state["finished"]])
    return len(compiled.tasks) - len(state["finished"])
