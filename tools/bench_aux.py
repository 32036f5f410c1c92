#!/usr/bin/env python
"""Secondary measurements for the BASELINE.json configs other than the headline (single GPU scale):

    python tools/bench_aux.py gemm32  [--n 16384]       # configs[4] family: fp32 GEMM program, 4096^2 tiles
    python tools/bench_aux.py tsqr    [--leaves 16]     # configs[3] family: (leaves*4096) x 4096 fp64 TSQR
    python tools/bench_aux.py chol    [--tiles 8]       # the Cholesky DAG on a larger tile grid
    python tools/bench_aux.py bdfac   [--tiles 4]       # alg_wrappers.bdfac (block bidiagonalisation), tiles x tiles grid
    python tools/bench_aux.py qr      [--tiles 4]       # alg_wrappers.qr (blocked QR program)
    python tools/bench_aux.py spill   [--tiles 4] [--budget-tiles 6]   # host-DRAM tier: copy rates, budgeted Cholesky

Each prints one JSON line.  Inputs are generated on the device and resident in HBM before timing;
programs are compiled before the clock starts (like bench.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.pop("NUMPYWREN_AMD_STORE", None)

from numpywren_amd import alg_wrappers, job_runner  # noqa: E402
from numpywren_amd import lambdapack as lp  # noqa: E402
from numpywren_amd.device import get_backend  # noqa: E402
from numpywren_amd.matrix import BigMatrix  # noqa: E402


STREAMS = 1
R_ONLY = True    # tsqr: drop the V / T factors nobody reads as they are stored (--keep-vt turns it off)
BATCH = None
SPILL_PLAN = True   # executor.spill_plan: the host-DRAM tier's victims / prefetches from the static DAG (False: LRU, on demand)
PREFETCH = None


def run(program, reclaim=True):
    program.config["executor"]["reclaim_intermediates"] = reclaim
    program.config["executor"]["drop_unread_outputs"] = reclaim and R_ONLY
    if BATCH is not None:
        program.config["executor"]["batch_tasks"] = BATCH
        program.config["executor"]["spill_batch_tasks"] = BATCH
    elif not SPILL_PLAN:
        program.config["executor"]["spill_batch_tasks"] = 32     # --lru: round 1's tier as it was (whole batches, on demand)
    program.config["executor"]["spill_plan"] = SPILL_PLAN
    if PREFETCH is not None:
        program.config["executor"]["spill_prefetch_tasks"] = PREFETCH
    program.start()
    job_runner.lambdapack_run(program, timeout=3600, pipeline_width=STREAMS)
    if program.program_status() != lp.PS.SUCCESS:
        raise SystemExit(f"failed: {program.exceptions}")


def timed(build, steps, warmup):
    be = get_backend()
    progs = [build() for _ in range(steps + warmup)]
    for p, meta in progs:
        p.program.tasks
        p._priorities()
    for i in range(warmup):
        p, meta = progs.pop(0)
        for m in meta["outputs"] + meta["intermediates"]:
            m.free()
        run(p)
    be.synchronize()
    t0 = time.time()
    for p, meta in progs:
        for m in meta["outputs"] + meta["intermediates"]:
            m.free()
        run(p)
    be.synchronize()
    return (time.time() - t0) / steps, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gemm32", "tsqr", "chol", "spill", "bdfac", "qr"])
    ap.add_argument("--budget-tiles", type=int, default=6)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--leaves", type=int, default=16)
    ap.add_argument("--tiles", type=int, default=8)
    ap.add_argument("--tile", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=1, help="HIP streams of the executor (pipeline_width)")
    ap.add_argument("--task-times", action="store_true", help="bdfac / qr: one extra run with per-kernel device times")
    ap.add_argument("--batch", type=int, default=None, help="executor.batch_tasks (ready tasks per batched launch)")
    ap.add_argument("--keep-vt", action="store_true", help="tsqr: keep the V / T factors (the reference's full output set)")
    ap.add_argument("--lru", action="store_true", help="spill: round 1's policy (LRU victims, restores on demand) instead of the DAG's plan")
    ap.add_argument("--prefetch", type=int, default=None, help="spill: executor.spill_prefetch_tasks")
    a = ap.parse_args()
    global STREAMS, BATCH, R_ONLY, SPILL_PLAN, PREFETCH
    SPILL_PLAN = not a.lru
    PREFETCH = a.prefetch
    STREAMS = a.streams
    R_ONLY = not a.keep_vt
    BATCH = a.batch
    os.environ.setdefault("NUMPYWREN_AMD_STREAMS", str(max(4, a.streams)))
    be = get_backend()
    b = a.tile
    if a.what == "gemm32":
        n = a.n
        nb = n // b
        A = BigMatrix("aux_A", shape=(n, n), shard_sizes=(b, b), dtype=np.float32)
        B = BigMatrix("aux_B", shape=(n, n), shard_sizes=(b, b), dtype=np.float32)
        for i in range(nb):
            for j in range(nb):
                A.put_tile(be.convert(be.fill_random((b, b), 11, i * b, j * b), np.float32), i, j)
                B.put_tile(be.convert(be.fill_random((b, b), 12, i * b, j * b), np.float32), i, j)
        dt, meta = timed(lambda: alg_wrappers.gemm(A, B), a.steps, a.warmup)
        C = meta["outputs"][0]
        # spot check one tile against a device-side fp64 product of the same operands
        ref = None
        for k in range(nb):
            t = be.gemm(be.as_f64(A.get_tile(0, k)), be.as_f64(B.get_tile(k, 1)), alpha=1.0, beta=1.0 if ref else 0.0, C=ref)
            ref = t
        diff = be.axpby(1.0, be.as_f64(C.get_tile(0, 1)), -1.0, ref)
        err = np.sqrt(be.sumsq(diff) / be.sumsq(ref))
        print(json.dumps({"what": f"{n}^2 fp32 GEMM program (alg_wrappers.gemm), {b}^2 tiles, {len(meta and alg_wrappers.gemm(A, B)[0].program.tasks)} tasks",
                          "ms": round(dt * 1e3, 2), "TFLOP/s": round(2 * n ** 3 / dt / 1e12, 2), "rel_err_tile_0_1": float(err),
                          "note": "fp32 MFMA products, fp64 add_matrices tree (reference promotion quirk)"}))
    elif a.what == "tsqr":
        m = a.leaves * b
        X = BigMatrix("aux_X", shape=(m, b), shard_sizes=(b, b))
        for j in range(a.leaves):
            X.put_tile(be.fill_random((b, b), 7, j * b, 0), j, 0)
        dt, meta = timed(lambda: alg_wrappers.tsqr(X), a.steps, a.warmup)
        levels = int(np.ceil(np.log2(a.leaves)))
        R = meta["outputs"][0].get_tile(levels, 0)
        # R^T R == A^T A
        G = None
        for j in range(a.leaves):
            t = X.get_tile(j, 0)
            G = be.gemm(t, t, True, False, alpha=1.0, beta=1.0 if G else 0.0, C=G)
        D = be.gemm(R, R, True, False, alpha=1.0, beta=-1.0, C=G)
        err = np.sqrt(be.sumsq(D) / be.sumsq(G))
        flops = 2 * m * b * b - 2 * b ** 3 / 3
        print(json.dumps({"what": f"{m} x {b} fp64 TSQR (alg_wrappers.tsqr), {a.leaves} leaves, {2 * a.leaves - 1} tasks",
                          "ms": round(dt * 1e3, 2), "TFLOP/s(2mn^2-2n^3/3)": round(flops / dt / 1e12, 3),
                          "rel_err_RtR": float(err), "streams": a.streams, "batch_tasks": a.batch or 32}))
    elif a.what in ("bdfac", "qr"):
        nt = a.tiles
        n = nt * b
        X = BigMatrix("aux_sq", shape=(n, n), shard_sizes=(b, b))
        for i in range(nt):
            for j in range(nt):
                X.put_tile(be.fill_random((b, b), 13, i * b, j * b), i, j)
        build = (lambda: alg_wrappers.bdfac(X)) if a.what == "bdfac" else (lambda: alg_wrappers.qr(X))
        dt, meta = timed(build, a.steps, a.warmup)
        ntasks = len(build()[0].program.tasks)
        out = {"what": f"{n}^2 fp64 alg_wrappers.{a.what}, {b}^2 tiles ({nt} x {nt}), {ntasks} tasks", "ms": round(dt * 1e3, 2),
               "batch_tasks": a.batch or 32, "streams": a.streams}
        if a.what == "qr":
            # Rs[0, 0, 0] is the R factor of the first block column's TSQR: R^T R = X0^T X0
            Rs = meta["outputs"][0]
            R = Rs.get_tile(0, 0, 0)
            G = None
            for i in range(nt):
                t = X.get_tile(i, 0)
                G = be.gemm(t, t, True, False, alpha=1.0, beta=1.0 if G else 0.0, C=G)
            D = be.gemm(R, R, True, False, alpha=1.0, beta=-1.0, C=G)
            out["rel_err_R00"] = float(np.sqrt(be.sumsq(D) / be.sumsq(G)))
            out["TFLOP/s(4n^3/3)"] = round(4 * n ** 3 / 3 / dt / 1e12, 2)
        else:
            out["TFLOP/s(8n^3/3)"] = round(8 * n ** 3 / 3 / dt / 1e12, 2)
        if a.task_times:
            # one more run with executor.task_timers: device time by kernel name (a task's bracket on its stream)
            prog, meta2 = build()
            prog.config["executor"]["task_timers"] = True
            for mat in meta2["outputs"] + meta2["intermediates"]:
                mat.free()
            run(prog)
            times = job_runner.collect_task_times(prog)
            out["task_ms"] = {k: [v["tasks"], round(v["ms"], 2)] for k, v in sorted(times.items(), key=lambda kv: -kv[1]["ms"])}
        print(json.dumps(out))
    elif a.what == "spill":
        from numpywren_amd import matrix
        # 1. the copies themselves: one tile out to pinned memory and back, on the spill stream
        t = be.fill_random((b, b), 3)
        be.synchronize()
        out, back, both = [], [], []
        t2nd = be.fill_random((b, b), 4)
        for _ in range(4):
            t0 = time.time()
            sp = be.spill_to_host(t)
            be.stream_sync(be.spill_stream())
            t1 = time.time()
            r = be.restore_from_host(sp)
            be.stream_sync(be.spill_stream(inbound=True))
            t2 = time.time()
            # both directions at once: one tile leaves while another comes back
            sp2 = be.spill_to_host(t2nd)
            r2 = be.restore_from_host(sp)
            be.stream_sync(be.spill_stream())
            be.stream_sync(be.spill_stream(inbound=True))
            t3 = time.time()
            out.append(t.nbytes / (t1 - t0) / 1e9)
            back.append(t.nbytes / (t2 - t1) / 1e9)
            both.append(2 * t.nbytes / (t3 - t2) / 1e9)
            del sp, r, sp2, r2
        # 2. the 16384^2-style Cholesky with all tiles resident and with a budget of a few tiles
        nt = a.tiles
        n = nt * b
        import bench

        res = {}
        for label, budget in (("resident", None), ("budget", a.budget_tiles * b * b * 8)):
            matrix.RESIDENCY.reset()
            matrix.RESIDENCY.set_budget(budget)
            s0, r0 = be.spilled_bytes_total, be.restored_bytes_total
            A = bench.build_input(be, nt, b, "aux_spill_" + label)
            times = []
            for rep in range(a.steps + a.warmup):
                program, meta = alg_wrappers.cholesky(A)
                program.program.tasks
                be.synchronize()
                t0 = time.time()
                run(program, reclaim=True)
                be.synchronize()
                times.append(time.time() - t0)
                last = meta
                if rep + 1 < a.steps + a.warmup:
                    for m in meta["outputs"] + meta["intermediates"]:
                        m.free()
            L = last["outputs"][0]
            res[label] = {"ms": round(1e3 * min(times[a.warmup:]), 2), "ms_all": [round(1e3 * t, 1) for t in times],
                          "GB_out_per_run": round((be.spilled_bytes_total - s0) / len(times) / 1e9, 2),
                          "GB_back_per_run": round((be.restored_bytes_total - r0) / len(times) / 1e9, 2), **matrix.RESIDENCY.stats()}
            res[label + "_L"] = [be.to_host(L.get_tile(nt - 1, j)) for j in range(nt)]
            A.free()
        same = all(np.array_equal(x, y) for x, y in zip(res.pop("resident_L"), res.pop("budget_L")))
        matrix.RESIDENCY.set_budget(None)
        print(json.dumps({"what": f"host-DRAM tier, {b}^2 fp64 tiles ({b * b * 8 >> 20} MiB)",
                          "d2h_pinned_GB/s": round(max(out), 1), "h2d_pinned_GB/s": round(max(back), 1),
                          "both_directions_GB/s": round(max(both), 1),
                          "cholesky": f"{n}^2, {nt}x{nt} tiles", "resident": res["resident"], "budget": res["budget"],
                          "budget_tiles": a.budget_tiles, "last_block_row_bitwise_equal": bool(same),
                          "pinned_bytes": be.pinned_bytes}))
    else:
        sys.argv = [sys.argv[0], "--tiles", str(a.tiles), "--tile", str(b), "--steps", str(a.steps), "--warmup", str(a.warmup),
                    "--no-cpu-baseline"]
        import bench
        bench.main()


if __name__ == "__main__":
    main()
