#!/usr/bin/env python
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

metric : achieved fp64 TFLOP/s of an N x N tiled Cholesky (4096^2 tiles) = (N^3 / 3) / wall, whole job.
step   : one complete factorisation (alg_wrappers.cholesky -> LambdaPACK DAG -> HIP-stream executor)
         of a synthetic SPD matrix whose tiles are already resident in HBM when the clock starts.
N = 1  : BASELINE.json configs[1], 16384 x 16384.  N > 1: tiles 2-D block-cyclic over the ranks,
         panel tiles exchanged over RCCL (numpywren_amd/dist.py); the problem grows with the GPU count
         (4 / 8 / 12 / 16 tiles per side for 1 / 2 / 4 / 8 GPUs; 8 GPUs = configs[2], 65536 x 65536).
Input  : tile (i, j) = X_i X_j^T + N * I[i == j], X = N x 128 counter-based standard normals generated on
         the device (SURVEY.md section 8d generator (ii)).  The reference experiment's own generator
         (x x^T + 20e12 N I) is NOT used for timing: with that diagonal shift every panel tile is < 1e-8,
         so the reference's allclose(x, 0) test in syrk skips every trailing update.
Extra objects on the JSON line (N = 1 only): `roofline` for the dominant kernel -- the syrk trailing
update, 2 * 4096^3 flop per launch, timed with HIP events on its launching stream inside the timed
steps -- and `cpu_baseline`: the oracle's tile Cholesky (NumPy / SciPy, the reference's kernel path
restated) on the host cores of this box.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense fp64 matrix peak (vendor spec; 74.6 measured by tools/devcheck)
TILE = 4096
SYRK_TRAFFIC_BYTES = 2.27e9    # measured, see profiles/r01_bench_rocprof_summary.md
TILES_PER_SIDE = {1: 4, 2: 8, 4: 12, 8: 16}


def build_input(be, nb, b, key, rank=0, world=1, owner=None):
    """Generate the SPD input directly in HBM, tile by tile (lower triangle only: the program never
    reads the upper one).  With several ranks each one materialises only the tiles it owns."""
    from numpywren_amd.matrix import BigMatrix
    n = nb * b
    X = BigMatrix(key, shape=(n, n), shard_sizes=(b, b), write_header=True)
    panels = {}

    def panel(i):
        if i not in panels:
            panels[i] = be.fill_random((b, 128), seed=2, row0=i * b, col0=0)
        return panels[i]

    for i in range(nb):
        for j in range(i + 1):
            if owner is not None and owner("I", (i, j)) != rank:
                continue
            t = be.gemm(panel(i), panel(j), False, True)
            if i == j:
                t = be.add_diag(t, float(n))
            X.put_tile(t, i, j)
    be.synchronize()
    return X


def cpu_baseline(b, budget_s=25.0):
    """Oracle (kind = "port") timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import npw_oracle as oracle
    rng = np.random.default_rng(0)
    probe = 2048
    a = rng.standard_normal((probe, probe))
    t0 = time.time()
    oracle.syrk(a, a, a)
    gflops = 2 * probe ** 3 / max(time.time() - t0, 1e-6) / 1e9
    executed = {nb: (nb * b ** 3 / 3 + nb * (nb - 1) / 2 * b ** 3 + sum((nb - i - 1) * (nb - i) / 2 for i in range(nb)) * 2 * b ** 3)
                for nb in (4, 3, 2, 1)}
    nb = next((k for k in (4, 3, 2, 1) if executed[k] / (gflops * 1e9 * 0.7) < budget_s), 1)
    n = nb * b
    X = rng.standard_normal((n, 128))
    tiles = {}
    for i in range(nb):
        for j in range(i + 1):
            t = X[i * b:(i + 1) * b] @ X[j * b:(j + 1) * b].T
            if i == j:
                t[np.diag_indices(b)] += n
            tiles[(i, j)] = t
    t0 = time.time()
    oracle.cholesky_tiles_inplace(tiles, nb)
    dt = time.time() - t0
    return {"value": round((n ** 3 / 3) / dt / 1e12, 4), "unit": "TFLOP/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle tile Cholesky (NumPy/SciPy BLAS+LAPACK, all host threads) of a {n} x {n} fp64 matrix, "
                      f"{b}^2 tiles, {dt:.2f} s; probe syrk {gflops:.0f} GFLOP/s",
            "blas": _blas_name()}


def _blas_name():
    try:
        cfg = np.show_config(mode="dicts")
        return cfg.get("Build Dependencies", {}).get("blas", {}).get("name", "unknown")
    except Exception:
        return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tiles", type=int, default=0, help="tiles per side (default: by GPU count)")
    ap.add_argument("--tile", type=int, default=TILE)
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams per rank (0 = 1 on one GPU, 3 with several: a task waiting for a tile in "
                         "transit must not block the tasks behind it)")
    ap.add_argument("--priority-stream", action="store_true", help="panel kernels on a high-priority stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.pop("NUMPYWREN_AMD_STORE", None)
    if args.streams <= 0:
        args.streams = 1 if world == 1 else 3

    from numpywren_amd import alg_wrappers, job_runner
    from numpywren_amd import lambdapack as lp
    from numpywren_amd.device import get_backend

    comm = None
    if world > 1 or os.environ.get("NUMPYWREN_AMD_FORCE_DIST"):   # the env var exercises the N > 1 code path on 1 GPU
        from numpywren_amd import dist
        comm = dist.init_process_group()   # RCCL over xGMI, one process per GPU
    be = get_backend()
    b = args.tile
    nb = args.tiles or TILES_PER_SIDE.get(args.gpus, 4 * args.gpus)
    n = nb * b

    owner = None
    if comm is not None:
        owner = comm.owner_fn(nb)
    X = build_input(be, nb, b, f"bench_chol_{n}_{b}", rank, world, owner)

    # programs are compiled before the clock starts (the reference reports compile_time separately,
    # alg_wrappers.py:20-24); scheduling, every kernel and every exchange are inside the timed region
    prebuilt = []
    for _ in range(args.warmup + args.steps):
        program, meta = alg_wrappers.cholesky(X)
        program.program.tasks
        program._priorities()
        prebuilt.append((program, meta))

    pending = []   # (program, meta) enqueued on the device, not yet waited for

    def settle():
        while pending:
            program, _ = pending.pop(0)
            program.wait()
            if program.program_status() != lp.PS.SUCCESS:
                raise SystemExit(f"cholesky failed: {program.exceptions}")
            program.free()

    def one_step():
        """One factorisation.  On one GPU the step is enqueued and the PREVIOUS one is waited for afterwards
        (program.wait() is where the reference's call sequence waits, too), so the host-side turnaround between two
        factorisations does not leave the GPU idle.  Steps do not overlap on the device (each waits for the previous
        one's completion events); every step is complete before the closing barrier."""
        program, meta = prebuilt.pop(0)
        for m in meta["outputs"] + meta["intermediates"]:
            m.free()
        program.config["executor"]["reclaim_intermediates"] = True
        program.config["executor"]["priority_stream"] = args.priority_stream
        program.start()
        if comm is None:
            # device-side order: this factorisation starts after the previous one has finished on the GPU (no overlap
            # of two steps); only the host runs ahead
            marks = pending[-1][0].completion_marks if pending else None
            job_runner.lambdapack_run(program, pipeline_width=args.streams, timeout=3600, wait=False, after=marks)
            settle()
            pending.append((program, meta))
        else:
            from numpywren_amd import dist
            dist.lambdapack_run_distributed(program, comm, pipeline_width=args.streams, timeout=3600)
            if program.program_status() != lp.PS.SUCCESS:
                raise SystemExit(f"cholesky failed: {program.exceptions}")
        return meta

    def barrier():
        settle()
        if comm is not None:
            comm.barrier()
        be.synchronize()

    for _ in range(args.warmup):
        one_step()
    if world == 1:
        be.enable_kernel_timers(("syrk", "syrk_sym", "trsm", "chol"))
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        meta = one_step()
    barrier()
    elapsed = time.time() - t0
    if comm is not None:
        elapsed = comm.max_over_ranks(elapsed)

    flops = n ** 3 / 3.0
    value = args.steps * flops / elapsed / 1e12
    line = {"metric": "achieved fp64 TFLOP/s, N x N tiled Cholesky (N^3/3 / wall)", "value": round(value, 3),
            "unit": "TFLOP/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n}x{n} fp64 Cholesky, {b}^2 tiles, {nb}x{nb} tile grid, alg_wrappers.cholesky "
                                   f"via LambdaPACK DAG ({nb*(nb+1)*(nb+2)//6} tasks)",
                       "n": n, "tile": b, "streams": args.streams,
                       "parallelism": "1 gpu" if world == 1 else f"{world} gpus, 2-D block-cyclic tiles, RCCL p2p panel exchange",
                       "pct_fp64_mfma_peak": round(100 * value / (FP64_MFMA_PEAK_TFLOPS * args.gpus), 2)}}
    if world == 1:
        times = be.collect_kernel_times()
        syrk = times.get("syrk", [])
        if syrk:
            avg_ms = float(np.mean(syrk))
            achieved = 2.0 * b ** 3 / (avg_ms * 1e-3) / 1e12
            line["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel<double,128,128,16,true,true,false,1> (kernels.syrk: S - X Y^T, 1024 workgroups)",
                                "achieved": round(achieved, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
                                # HBM-side bytes per launch from the PMC passes of profiles/r01_bench_rocprof_summary.md
                                # (2 x FETCH_SIZE + WRITE_SIZE); only meaningful for the 4096^2 tile it was measured on
                                "traffic": SYRK_TRAFFIC_BYTES if b == TILE else None,
                                "traffic_unit": "B/launch (PMC, separate passes; algorithmic 5.37e8)",
                                "launches": len(syrk), "avg_ms": round(avg_ms, 4),
                                "algorithmic_flop_per_launch": 2 * b ** 3}
            line["kernel_ms"] = {k: round(float(np.mean(v)), 4) for k, v in times.items() if v}
        # parity guard at full size: || A - L L^T ||_F / || A ||_F on the diagonal-block row 0..1 (cheap, device side)
        O = meta["outputs"][0]
        L00, L10, L11 = O.get_tile(0, 0), O.get_tile(1, 0), O.get_tile(1, 1)
        A11 = X.get_tile(1, 1)
        r = be.gemm(L10, L10, False, True, alpha=-1.0, beta=1.0, C=A11)
        r = be.gemm(L11, L11, False, True, alpha=-1.0, beta=1.0, C=r)
        line["config"]["residual_tile_1_1"] = float(np.sqrt(be.sumsq(r) / be.sumsq(A11)))
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(b)
    if rank == 0:
        print(json.dumps(line))
    if comm is not None:
        comm.shutdown()


if __name__ == "__main__":
    main()
