#!/usr/bin/env python
"""What the distributed walk (numpywren_amd/dist.py: lambdapack_run_distributed) costs on the HOST, on the real backend
(VERDICT r4 item 2).  One GPU is enough for both measurements:

  world1   the N > 1 code path with a world of one (RcclTransport, every task owned here): the whole walk's host time,
           split by dist.py's own clock reads (diag["host_split_ms"]) -- dequeue, look-ups, run:<kernel>, exchange plan,
           post_op -- once with the default run-ahead bound and once with a small one (a wait hidden inside a HIP call
           moves into host_blocked_ms when the host is kept close behind the device);
  pretend  this process plays rank R of a job of W ranks (default 0 of 8, the 2 x 4 grid of bench.py --gpus 8) on the same
           65536^2 matrix: it executes only the tasks rank R owns, its sends are discarded, and every tile it would receive
           is handed over from a complete factorisation made beforehand on this GPU (only factor tiles ever travel in the
           Cholesky program).  The host cost of a 1/W owner is then measured, not inferred.

--profile N prints the N most expensive functions (cProfile, own time) of the world-of-one walk.
Writes one JSON object to stdout; tools/predict_scaling.py --host-split <file> reads it."""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NUMPYWREN_AMD_FORCE_DIST", "1")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_PORT", "29677")

import bench  # noqa: E402
from numpywren_amd import alg_wrappers, dist  # noqa: E402
from numpywren_amd import lambdapack as lp  # noqa: E402
from numpywren_amd.device import get_backend  # noqa: E402


class HandOverTransport(object):
    """Stand-in transport of the pretended rank: sends vanish, a receive hands over the tile of a finished factorisation."""
    name = "pretend"
    wants_key = True
    diag = False

    def __init__(self, source):
        self.source = source      # {matrix name: BigMatrix holding every tile}
        self.sent = self.received = 0

    def begin_group(self):
        pass

    def end_group(self):
        pass

    def abort_group(self):
        pass

    def send(self, tile, dsts):
        self.sent += len(dsts)

    def recv(self, src, meta, key):
        self.received += 1
        return self.source[key[0]].get_tile(*key[1])

    def exchange_ms(self):
        return 0.0

    def flush(self):
        pass

    def close(self):
        pass


class PretendComm(dist.Comm):
    """Rank `rank` of `world`, alone: control collectives return the local value."""

    def __init__(self, rank, world, transport):
        dist.Comm.__init__(self, rank, world, transport, None)

    def barrier(self):
        pass

    def max_over_ranks(self, value):
        return float(value)

    def shutdown(self):
        pass


def run_once(program, meta, comm, streams, max_inflight, profile=0):
    for m in meta["outputs"] + meta["intermediates"]:
        m.free()
    program.config["executor"]["reclaim_intermediates"] = True
    program.start()
    prof = cProfile.Profile() if profile else None
    t0 = time.time()
    if prof:
        prof.enable()
    res = dist.lambdapack_run_distributed(program, comm, pipeline_width=streams, timeout=3600, max_inflight=max_inflight)
    if prof:
        prof.disable()
    wall = time.time() - t0
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    d = dict(res["diag"], wall_ms=round(1e3 * wall, 3), max_inflight=max_inflight)
    if prof:
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(profile)
        d["profile_top"] = buf.getvalue().splitlines()[-(profile + 3):]
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=16)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--pretend-world", type=int, default=8)
    ap.add_argument("--pretend-rank", type=int, default=0)
    ap.add_argument("--profile", type=int, default=0)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    be = get_backend()
    comm = dist.init_process_group()
    nb, b = args.tiles, args.tile
    X = bench.build_input(be, nb, b, f"split_chol_{nb}_{b}")
    out = {"n": nb * b, "tile": b, "tasks": nb * (nb + 1) * (nb + 2) // 6, "streams": args.streams, "world1": [], "pretend": []}
    build = lambda: alg_wrappers.cholesky(X)
    progs = bench._prebuild(build, 2 + 2 * args.reps + 1 + args.reps)
    run_once(*progs.pop(), comm, args.streams, 64)                      # warm-up (allocator, code objects)
    for inflight in (64, 4):
        for _ in range(args.reps):
            out["world1"].append(run_once(*progs.pop(), comm, args.streams, inflight))
    if args.profile:
        out["world1_profiled"] = run_once(*progs.pop(), comm, args.streams, 64, profile=args.profile)
    # the complete factor, kept: what the pretended rank's receives hand over
    program, meta = progs.pop()
    run_once(program, meta, comm, args.streams, 64)
    full = {"O": meta["outputs"][0]}
    keep = {tuple(i): full["O"].get_tile(*i) for i in full["O"].block_idxs_exist}
    pcomm = PretendComm(args.pretend_rank, args.pretend_world, HandOverTransport(None))

    class Kept(object):
        def get_tile(self, *idx):
            return keep[tuple(idx)]
    pcomm.transport.source = {"O": Kept()}
    for _ in range(args.reps):
        program, meta = bench._prebuild(build, 1)[0]
        for m in meta["outputs"] + meta["intermediates"]:
            m.free()
        d = run_once(program, meta, pcomm, args.streams, 64)
        d["sends_discarded"], d["tiles_handed_over"] = pcomm.transport.sent, pcomm.transport.received
        pcomm.transport.sent = pcomm.transport.received = 0
        out["pretend"].append(d)
    out["pretend_rank"], out["pretend_world"] = args.pretend_rank, args.pretend_world
    print(json.dumps(out))
    comm.shutdown()


if __name__ == "__main__":
    main()
