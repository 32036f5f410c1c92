#!/usr/bin/env python
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Either form works for N > 1: without a launcher's environment (no WORLD_SIZE) `--gpus N` starts its own N ranks -- one
process per GPU under torch.distributed.run on 127.0.0.1, like the reference's drivers launch their own workers
(experiments/cholesky_experiment.py:170) -- and a line is only ever printed when exactly N ranks joined: `n_gpus` is
the size of the job that ran, `config.ranks_joined` / `config.devices` / `config.rccl_nranks` are the evidence.
`--dry-run` stops after the rendezvous and prints who joined (no GPU work).

metric : achieved fp64 TFLOP/s of an N x N tiled Cholesky (4096^2 tiles) = (N^3 / 3) / wall, whole job.
step   : one complete factorisation (alg_wrappers.cholesky -> LambdaPACK DAG -> HIP-stream executor)
         of a synthetic SPD matrix whose tiles are already resident in HBM when the clock starts.
N = 1  : BASELINE.json configs[1], 16384 x 16384 (+ a `north_star` object: configs[2]'s 65536 x 65536 matrix on
         this ONE GPU, 3 steps).  N > 1: STRONG scaling of the 65536 x 65536 matrix (configs[2] at N = 8): tiles 2-D
         block-cyclic over the ranks, panel tiles pushed point-to-point over RCCL / xGMI by libnpw_hip.so's
         npw_comm_* layer (numpywren_amd/dist.py).  --workload tsqr / gemm32 time configs[3] / configs[4] the same way.
Input  : tile (i, j) = X_i X_j^T + N * I[i == j], X = N x 128 counter-based standard normals generated on
         the device (SURVEY.md section 8d generator (ii)).  The reference experiment's own generator
         (x x^T + 20e12 N I) is NOT used for timing: with that diagonal shift every panel tile is < 1e-8,
         so the reference's allclose(x, 0) test in syrk skips every trailing update.
Extra objects on the JSON line (N = 1 only): `roofline` for the dominant kernel -- the syrk trailing
update, 2 * 4096^3 flop per launch, timed with HIP events on its launching stream inside the timed
steps -- and `cpu_baseline`: the oracle's tile Cholesky (NumPy / SciPy, the reference's kernel path
restated) on the host cores of this box, at the best BLAS thread count of a sweep and at 1 thread.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E nominal (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming kernel achieves
HBM_ACHIEVABLE_GBPS = 6300.0
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (vendor spec)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense fp64 matrix peak (vendor spec; 74.6 measured by tools/devcheck)
TILE = 4096
# 65536^2 on ONE GPU: the anchor of the N > 1 strong-scaling lines, and the curve DESIGN.md section 6 predicts from the
# measured kernel times (never a measured value: no multi-GPU node has been available to the builder)


def _predicted_scaling(workload):
    """The predicted 1 / 2 / 4 / 8-GPU figures of `workload` from profiles/predicted_scaling.json -- written by
    tools/predict_scaling.py from measured single-GPU kernel times; the ONE table DESIGN.md and BASELINE.md quote too.
    Never a measured value."""
    try:
        with open(os.path.join(ROOT, "profiles", "predicted_scaling.json")) as f:
            table = json.load(f)
        return {"source": "profiles/predicted_scaling.json (tools/predict_scaling.py: a simulation, not a measurement)",
                **table["workloads"][workload]}
    except Exception:
        return None


def _predicted_at_link(workload, world, link_gbs):
    """The model behind profiles/predicted_scaling.json (tools/predict_scaling.py) re-evaluated for THIS world size at the link
    rate link_calibration() just measured on this node -- beside the table's own figure, which assumes 64 GB/s per direction.
    Still a simulation: what it adds is that the one free parameter nobody had measured is now this node's."""
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("npw_predict_scaling", os.path.join(ROOT, "tools", "predict_scaling.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        row = mod.predict(workload, world, float(link_gbs))
        return {"link_GBps_per_direction": round(float(link_gbs), 2), "gpus": world, "ms": row["ms"], "tflops": row["tflops"],
                "source": "tools/predict_scaling.py predict() at the measured median link rate (a simulation, not a measurement)"}
    except Exception as exc:       # the model must never cost a bench line
        return {"error": repr(exc)}


def _syrk_traffic():
    """(bytes per tile update, source) of the trailing-update kernel from the newest profiles/r*_bench_pmc.json: separate
    rocprofv3 --pmc passes of this command, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / tiles per dispatch (gfx950 reports wide
    coalesced reads at half their size: MI355X_MICROARCH.md).  Not measured in this process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                b_ = json.load(f)["bench"]
            per_tile = (2.0 * b_["FETCH_SIZE"] + b_["WRITE_SIZE"]) * 1024.0 / b_.get("tiles_per_dispatch", 1)
            return per_tile, "profiles/" + os.path.basename(path)
        except Exception:
            continue
    return None, None


def _free_port():
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def launch_ranks(gpus, argv):
    """`--gpus N` (N > 1) outside a launcher: become `python -m torch.distributed.run` with N ranks of this script on
    this node (one process per GPU; LOCAL_RANK picks the HIP device).  Replaces the current process, so the exit code
    and rank 0's JSON line are the job's."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes here
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def job_identity(comm, world, rank):
    """Who joined: every rank's (rank, host, pid, LOCAL_RANK, PCI bus id of its HIP device or None) over the control
    group, plus the size RCCL itself reports for the payload communicator."""
    import socket
    mine = {"rank": rank, "host": socket.gethostname(), "pid": os.getpid(), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
            "device": None, "rccl_nranks": None, "visible_devices": None,
            # what narrows the device list of this process, if anything (a launcher that sets one GPU per rank AND leaves
            # LOCAL_RANK counting up is fine -- the modulo below -- two ranks with the same list and the same LOCAL_RANK are not)
            "env": {k: os.environ[k] for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if k in os.environ}}
    try:
        from numpywren_amd.device import hip_available
        if hip_available():
            import ctypes
            from numpywren_amd import _ffi
            n = ctypes.c_int(0)
            _ffi.lib().npw_device_count(ctypes.byref(n))
            buf = ctypes.create_string_buffer(64)
            mine["visible_devices"] = n.value
            if n.value > 0 and _ffi.lib().npw_device_pci_bus_id(mine["local_rank"] % n.value, buf, 64) == 0:
                mine["device"] = buf.value.decode()
    except Exception:
        pass
    tr = getattr(comm, "transport", None)
    if getattr(tr, "handle", None) and hasattr(tr, "lib"):
        import ctypes
        w = ctypes.c_int(0)
        if tr.lib.npw_comm_info(tr.handle, None, ctypes.byref(w), None) == 0:
            mine["rccl_nranks"] = w.value
    joined = [None] * world
    if world > 1:
        comm.dist.all_gather_object(joined, mine)
    else:
        joined = [mine]
    return joined


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


def _blas_threads(n):
    """Context manager limiting the BLAS thread pool (threadpoolctl when present, else a no-op)."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=int(n))
    except Exception:
        import contextlib
        return contextlib.nullcontext()


def _oracle_tile_cholesky(oracle, b, nb, rng):
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
    return n, time.time() - t0


def cpu_baseline(b, budget_s=12.0):
    """Oracle (kind = "port": the reference's NumPy / SciPy kernel path restated, oracle/npw_oracle.py) timed on this
    box's host cores on a bounded sample of the same workload.  A fair one: the BLAS thread count is swept on a
    2048^3 syrk probe and the tile Cholesky runs at the best setting AND at 1 thread -- the reference's per-worker
    default (experiments/cholesky_experiment.py:66,367) -- each sample sized to ~`budget_s` seconds; config 1's
    4096^2 single-tile gemm is timed at both settings as well."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import npw_oracle as oracle
    rng = np.random.default_rng(0)
    best, sweep = _best_blas_threads(oracle, rng)

    def executed(nb):   # flops the tile DAG executes (chol b^3/3, trsm b^3, syrk 2 b^3)
        return (nb * b ** 3 / 3 + nb * (nb - 1) / 2 * b ** 3 + sum((nb - i - 1) * (nb - i) / 2 for i in range(nb)) * 2 * b ** 3)

    out = {"unit": "TFLOP/s", "kind": "port", "blas": _blas_name() + " (OpenBLAS builds of NumPy / SciPy; MKL is not in this image)",
           "probe_syrk_2048_gflops_by_threads": sweep}
    for label, th in (("best", best), ("one_thread", 1)):
        # the best setting always runs configs[1]'s own 16384^2 matrix (10 - 25 s on these hosts), so that `value` means
        # the same on every box of the pool; the 1-thread sample is sized to the budget
        nb = 4 if label == "best" else next((k for k in (4, 3, 2, 1) if executed(k) / (sweep[th] * 1e9 * 0.7) < budget_s), 1)
        with _blas_threads(th):
            n, dt = _oracle_tile_cholesky(oracle, b, nb, rng)
        out[label] = {"threads": th, "tflops": round((n ** 3 / 3) / dt / 1e12, 4), "n": n, "seconds": round(dt, 2)}
    # config 1: one 4096^2 fp64 kernels.gemm on the CPU path (median of 5 at the best setting, of 3 at one thread)
    A, B_ = rng.standard_normal((b, b)), rng.standard_normal((b, b))
    gem = {}
    for label, th, reps in (("best", best, 5), ("one_thread", 1, 3)):
        with _blas_threads(th):
            if label == "best":
                oracle.gemm(A, B_)
            ts = []
            for _ in range(reps):
                t0 = time.time()
                oracle.gemm(A, B_)
                ts.append(time.time() - t0)
        gem[label] = {"threads": th, "median_s": round(float(np.median(ts)), 4),
                      "tflops": round(2 * b ** 3 / float(np.median(ts)) / 1e12, 4)}
    out["config1_gemm_4096"] = gem
    out["value"] = out["best"]["tflops"]
    out["cores"] = out["best"]["threads"]
    out["sample"] = (f"oracle tile Cholesky (NumPy/SciPy BLAS+LAPACK) of a {out['best']['n']}^2 fp64 matrix, {b}^2 tiles, at the "
                     f"best BLAS thread count of the sweep ({best}; {out['best']['seconds']} s) and of a "
                     f"{out['one_thread']['n']}^2 one at 1 thread, the reference worker's default "
                     f"({out['one_thread']['seconds']} s)")
    return out


def _blas_name():
    try:
        cfg = np.show_config(mode="dicts")
        return cfg.get("Build Dependencies", {}).get("blas", {}).get("name", "unknown")
    except Exception:
        return "unknown"


def _best_blas_threads(oracle, rng):
    """(threads, {threads: GFLOP/s}) of the 2048^3 syrk probe -- the sweep cpu_baseline() runs for the Cholesky line."""
    ncpu = os.cpu_count() or 1
    probe = 2048
    a = rng.standard_normal((probe, probe))
    sweep = {}
    for th in sorted({1, 32, 64, 128, ncpu} & set(range(1, ncpu + 1)) | {1, ncpu}):
        with _blas_threads(th):
            oracle.syrk(a, a, a)
            t0 = time.time()
            oracle.syrk(a, a, a)
            sweep[th] = round(2 * probe ** 3 / max(time.time() - t0, 1e-6) / 1e9, 1)
    return max(sweep, key=sweep.get), sweep


def cpu_baseline_tsqr(b, leaves_total, sample_leaves=2):
    """the oracle's TSQR (oracle.tsqr: LAPACK DGEQRT through SciPy on every leaf and tree node, the
    reference's fast_qr restated) on a `sample_leaves`-leaf, b-wide slice of the same input, all host cores, scaled to the
    full problem by algorithmic flops (2 m n^2 - 2 n^3 / 3).  Two leaves and their tree node, not the survey's 16 - 32: SciPy's DGEQRT
    takes 3 - 5 s per 4096^2 leaf and 6 - 10 s per 8192 x 4096 node on these hosts (8 leaves: 65 s, 4 leaves: 28 - 40 s), and the
    contract bounds the sample at 10 - 30 s.  kind = "port": the reference's own LAPACK module is an f2py
    build it downloads at run time (kernels.py:12-40) and is not in its tree."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import npw_oracle as oracle
    rng = np.random.default_rng(7)
    best, sweep = _best_blas_threads(oracle, rng)
    m = sample_leaves * b
    X = rng.standard_normal((m, b))
    with _blas_threads(best):
        t0 = time.time()
        oracle.tsqr(X, b)
        dt = time.time() - t0
    flops = 2.0 * m * b * b - 2.0 * b ** 3 / 3
    full = 2.0 * leaves_total * b * b * b - 2.0 * b ** 3 / 3
    tf = flops / dt / 1e12
    return {"value": round(tf, 4), "unit": "TFLOP/s", "cores": best, "kind": "port",
            "blas": _blas_name() + " (OpenBLAS build of SciPy; MKL is not in this image)", "probe_syrk_2048_gflops_by_threads": sweep,
            "seconds": round(dt, 2), "extrapolated_full_problem_s": round(full / (tf * 1e12), 1),
            "sample": f"oracle TSQR (SciPy LAPACK DGEQRT on {sample_leaves} leaves + {sample_leaves - 1} tree nodes) of a {m}x{b} fp64 "
                      f"slice at {best} BLAS threads, {round(dt, 1)} s; the {leaves_total}-leaf problem is extrapolated by "
                      f"algorithmic flops, not run"}


def cpu_baseline_gemm32(b, nb):
    """one oracle 4096^3 sgemm (kernels.gemm restated: A.dot(B) on float32 tiles), all host cores, median
    of 3; the nb^3 products of the program (and nothing for its fp64 add tree) extrapolated."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import npw_oracle as oracle
    rng = np.random.default_rng(11)
    best, sweep = _best_blas_threads(oracle, rng)
    A = rng.standard_normal((b, b)).astype(np.float32)
    B_ = rng.standard_normal((b, b)).astype(np.float32)
    with _blas_threads(best):
        oracle.gemm(A, B_)
        ts = []
        for _ in range(3):
            t0 = time.time()
            oracle.gemm(A, B_)
            ts.append(time.time() - t0)
    med = float(np.median(ts))
    tf = 2.0 * b ** 3 / med / 1e12
    return {"value": round(tf, 4), "unit": "TFLOP/s", "cores": best, "kind": "port",
            "blas": _blas_name() + " (OpenBLAS build of NumPy; MKL is not in this image)", "probe_syrk_2048_gflops_by_threads": sweep,
            "median_s": round(med, 4), "extrapolated_full_problem_s": round(nb ** 3 * med, 1),
            "sample": f"one oracle {b}^3 fp32 gemm (NumPy sgemm) at {best} BLAS threads, median of 3; the program's {nb ** 3} "
                      f"products are extrapolated (its fp64 add_matrices tree not counted), not run"}


def _profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


def _prebuild(build, count):
    """Compile `count` programs before the clock starts (the reference reports compile_time separately,
    alg_wrappers.py:20-24); scheduling, every kernel and every exchange stay inside the timed region."""
    out = []
    for _ in range(count):
        program, meta = build()
        program.program.tasks
        program._priorities()
        out.append((program, meta))
    return out


class Runner(object):
    """Times K runs of one alg_wrappers program after W warm-up runs, bracketed by a barrier + device synchronise."""

    def __init__(self, be, comm, streams, priority_stream=False, r_only=False):
        self.be, self.comm, self.streams, self.priority_stream = be, comm, streams, priority_stream
        self.r_only = r_only    # TSQR: drop the V / T factors no task reads as they are stored (executor.drop_unread_outputs)
        self.fuse = False       # GEMM program: executor.fuse_gemm_reduction (the accumulate-in-place mode)
        self.task_timers = False  # N > 1: executor.task_timers in the extra diagnostic step after the timed region
        self.last_dist = None   # what dist.lambdapack_run_distributed returned for the last step (its "diag" entry)
        self.pending = []   # (program, meta) enqueued on the device, not yet waited for
        self.step_ms = []   # the last timed() call's per-step durations

    def settle(self):
        from numpywren_amd import lambdapack as lp
        while self.pending:
            program, _ = self.pending.pop(0)
            program.wait()
            if program.program_status() != lp.PS.SUCCESS:
                raise SystemExit(f"program failed: {program.exceptions}")
            program.free()

    def one_step(self, program, meta):
        """One run.  On one GPU the step is enqueued and the PREVIOUS one is waited for afterwards (program.wait() is
        where the reference's call sequence waits, too), so the host-side turnaround between two runs does not leave
        the GPU idle.  Steps do not overlap on the device (each waits for the previous one's completion events);
        every step is complete before the closing barrier."""
        from numpywren_amd import job_runner
        from numpywren_amd import lambdapack as lp
        for m in meta["outputs"] + meta["intermediates"]:
            m.free()
        program.config["executor"]["reclaim_intermediates"] = True
        program.config["executor"]["drop_unread_outputs"] = self.r_only
        program.config["executor"]["fuse_gemm_reduction"] = self.fuse
        program.config["executor"]["priority_stream"] = self.priority_stream
        program.start()
        if self.comm is None:
            marks = self.pending[-1][0].completion_marks if self.pending else None
            job_runner.lambdapack_run(program, pipeline_width=self.streams, timeout=3600, wait=False, after=marks)
            self.settle()
            self.pending.append((program, meta))
            if os.environ.get("NPW_BENCH_DEBUG"):
                be = self.be
                print(f"[bench] allocated {be.allocated_bytes / 2**30:.1f} GiB, pooled {be.pooled_bytes / 2**30:.1f}, peak "
                      f"{be.peak_bytes / 2**30:.1f}, spilled {be.spilled_bytes_total / 2**30:.1f}, restored "
                      f"{be.restored_bytes_total / 2**30:.1f}, allocator syncs {getattr(be, 'alloc_syncs', 0)}", file=sys.stderr)
                print("        pool: " + ", ".join(f"{k / 2**20:.0f}MiB x{len(v)}" for k, v in sorted(be._free.items()) if v and k >= 2**26),
                      "| pending:", len(be._pending), file=sys.stderr)
        else:
            from numpywren_amd import dist
            program.config["executor"]["task_timers"] = self.task_timers
            self.last_dist = dist.lambdapack_run_distributed(program, self.comm, pipeline_width=self.streams, timeout=3600)
            if program.program_status() != lp.PS.SUCCESS:
                raise SystemExit(f"program failed: {program.exceptions}")
        return meta

    def barrier(self):
        self.settle()
        if self.comm is not None:
            self.comm.barrier()
        self.be.synchronize()

    def _mark_step(self, program):
        """A timing event behind everything the step just enqueued (one GPU: steps are pipelined on the device, so the
        host clock does not see where one ends; the event costs ~4 us of stream time per step)."""
        be = self.be
        s0 = (getattr(be, "bulk_streams", None) or be.streams)[0]
        for ev in (getattr(program, "completion_marks", None) or []) if program is not None else []:
            be.wait_event(s0, ev)
        ev = be.new_event(timing=True)
        be.record(ev, s0)
        return ev

    def timed(self, build, steps, warmup, timers=None):
        """(elapsed seconds of `steps` runs between two barriers, the last run's meta).  self.step_ms afterwards: each
        timed step's own duration -- on one GPU from timing events between the steps' completion marks, with several
        ranks from the host clock around each (synchronous) step, maximum over the ranks."""
        progs = _prebuild(build, steps + warmup)
        for _ in range(warmup):
            self.one_step(*progs.pop(0))
        if timers and self.comm is None:
            self.barrier()
            self.be.enable_kernel_timers(timers)
        self.barrier()
        device_marks = self.comm is None and hasattr(self.be, "new_event")
        marks = [self._mark_step(None)] if device_marks else []
        host = []
        t0 = time.time()
        for _ in range(steps):
            t1 = time.time()
            program = progs[0][0]
            meta = self.one_step(*progs.pop(0))
            host.append(time.time() - t1)
            if device_marks:
                marks.append(self._mark_step(program))
        self.barrier()
        elapsed = time.time() - t0
        if self.comm is not None:
            elapsed = self.comm.max_over_ranks(elapsed)
            self.step_ms = [round(1e3 * self.comm.max_over_ranks(h), 3) for h in host]
        elif device_marks:
            self.step_ms = [round(self.be.elapsed_ms(a, b), 3) for a, b in zip(marks[:-1], marks[1:])]
        else:
            self.step_ms = [round(1e3 * h, 3) for h in host]
        return elapsed, meta


def step_stats(step_ms, mean_ms, what="ms_per_step"):
    """`step_ms` + median + outliers (> 1.5 x the median) for a JSON line whose `ms_per_step` stays the MEAN over the
    timed bracket (what the driver's own clock checks); a stalled step -- rocm-smi polling on the node has been seen to
    stall single repetitions by 0.25 - 4 s (profiles/r04_qr_tsqr.md) -- then shows on the line instead of silently
    halving the figure.  Warns on stderr when mean and median differ by more than 2 %."""
    if not step_ms:
        return {}
    med = float(np.median(step_ms))
    out = {"step_ms": list(step_ms), what + "_median": round(med, 3),
           "outliers": [i for i, t in enumerate(step_ms) if t > 1.5 * med]}
    if med > 0 and abs(mean_ms - med) > 0.02 * med:
        print(f"[bench] warning: {what} mean {mean_ms:.3f} ms and median {med:.3f} ms differ by "
              f"{100 * (mean_ms - med) / med:+.1f} % (step_ms {step_ms}); outliers at steps {out['outliers']}", file=sys.stderr)
    return out


def cholesky_residual(be, X, O, nb, full=False):
    """|| A - L L^T ||_F / || A ||_F on the device: all tiles (full=True) or the tile (1, 1).  A PROPERTY check computed with
    this build's own GEMM (be.gemm) -- not an independent product: that GEMM is checked against the oracle at 4096^2 in
    tests/test_tile4096_gpu.py, and the factor itself against np.linalg.cholesky / the oracle at the sizes those finish."""
    num = den = 0.0
    todo = [(i, j) for i in range(nb) for j in range(i + 1)] if full else [(1, 1) if nb > 1 else (0, 0)]
    for i, j in todo:
        a = X.get_tile(i, j)
        r = a
        for k in range(j + 1):
            r = be.gemm(O.get_tile(i, k), O.get_tile(j, k), False, True, alpha=-1.0, beta=1.0, C=r)
        w = 1.0 if i == j else 2.0
        num += w * be.sumsq(r)
        den += w * be.sumsq(a)
    return float(np.sqrt(num / den))


def one_gpu_anchor(be, comm, rank, what, make_input, build, flops, r_only=False):
    """N > 1 lines: the SAME problem on one GPU of this box -- rank 0 alone with the plain one-GPU executor, before the
    communicator exists (the other ranks wait at the barrier) -- so that a scaling curve has its own measured N = 1 point."""
    anchor = None
    if rank == 0:
        solo = Runner(be, None, 1, r_only=r_only)
        mats = make_input()
        ea, ma = solo.timed(lambda: build(*mats), 2, 1)
        anchor = {"what": what + " on one GPU (rank 0, before the timed multi-GPU steps)", "steps": 2,
                  "ms_per_step": round(ea / 2 * 1e3, 3), "tflops": round(2 * flops / ea / 1e12, 3), "step_ms": solo.step_ms}
        for m_ in list(mats) + ma["outputs"] + ma["intermediates"]:
            m_.free()
        del mats, ma, solo
        if hasattr(be, "trim"):
            be.trim()
    comm.barrier()
    return anchor


def tsqr_roofline(times, b, r_only):
    """The TSQR line's roofline: the hot path is the C-ABI call npw_dgeqrt_batched (the leaves, batches of 32) and
    npw_dtpqrt_batched (the tree nodes) -- each a sequence of launches on four streams inside the library, bracketed here by
    HIP events on the caller's stream (the call joins its helper streams before it returns).  Per tile: algorithmic
    Householder flops (4/3 b^3 for a leaf, the same count for a 2b x b node as LAPACK's dense DGEQRT would spend: 10/3 b^3 --
    the structured kernel executes about a third of that, so a node's `frac` can look generous; the leaf is what is quoted)
    over the measured time per tile.  Both bounds are stated: the MFMA time of those flops at the fp64 peak and the HBM time
    of the PMC-measured bytes of one batch of 32 at 6.3 TB/s (the achievable rate of MI355X_MICROARCH.md); the larger is `bound`."""
    leaf = times.get("geqrt_batched", [])
    node = times.get("tpqrt_batched", [])
    if not leaf:
        return None
    avg_ms = float(np.mean(leaf))
    flop = 4.0 * b ** 3 / 3
    tflops = flop / (avg_ms * 1e-3) / 1e12
    pmc = _profile_json("r06_qr32_hbm_bytes.json") or {}
    key = "r_only" if r_only else "with_t"
    bytes32 = (pmc.get(key) or {}).get("bytes_per_batch_of_32") if b == TILE else None
    t_mfma = 32 * flop / (FP64_MFMA_PEAK_TFLOPS * 1e12) * 1e3
    t_hbm = bytes32 / (HBM_ACHIEVABLE_GBPS * 1e9) * 1e3 if bytes32 else None
    kernel = ("npw_dgeqrt_batched x32 (kernels.qr_factor on 32 leaves of 4096^2: panel chain + three levels of block reflectors; "
              "dominant device kernels by rocprof share: gemm_kernel<double,128,128,16,true,true,false,0,2> -- the rank-256 far update -- "
              "and gemm_kernel<double,128,256,16,false,false,false,4,4> -- its X^T = W2^T V product; profiles/r06_qr_batched32*_kernel_stats.csv)")
    mfma_view = {"achieved": round(tflops, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / FP64_MFMA_PEAK_TFLOPS, 4),
                 "bound_ms_per_batch_of_32": round(t_mfma, 2), "algorithmic_flop_per_tile": flop}
    common = {"kernel": kernel, "launches": len(leaf), "avg_ms": round(avg_ms, 4), "avg_ms_is": "per leaf tile (a call of 32 tiles / 32)",
              "algorithmic_flop_per_launch": flop, "mfma": mfma_view,
              "traffic": bytes32,
              "traffic_unit": "B per batch of 32 leaves (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, separate passes; "
                              "profiles/r06_qr32_hbm_bytes.json), not this run; compulsory (read A, write R" + ("" if r_only else ", V, T")
                              + "): " + ("8.6e9" if r_only else "1.72e10")}
    # `bound` follows the contract: by its ALGORITHMIC work (170 flop per compulsory byte) the factorisation is MFMA-bound, `achieved`
    # is algorithmic flop over measured time.  Beside it the other lower bound VERDICT r5 asked for: the HBM time of the bytes
    # the batch really moves (PMC) at the achievable 6.3 TB/s -- when that exceeds the MFMA time, wasted traffic is what
    # stands between the batch and its roofline, and `larger_lower_bound` says so.
    out = dict(common, bound="mfma", achieved=mfma_view["achieved"], peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=mfma_view["frac"],
               mfma_bound_ms_per_batch_of_32=round(t_mfma, 2), hbm_bound_ms_per_batch_of_32=round(t_hbm, 2) if t_hbm else None,
               measured_ms_per_batch_of_32=round(32 * avg_ms, 2))
    if t_hbm is not None:
        gbps = bytes32 / (32 * avg_ms * 1e-3) / 1e9
        out["hbm"] = {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                      "achievable_GBps": HBM_ACHIEVABLE_GBPS, "what": "MEASURED (PMC) bytes per batch over the measured batch time"}
        out["larger_lower_bound"] = ("hbm: %.1f ms for the measured traffic at 6.3 TB/s against %.1f ms of MFMA time" % (t_hbm, t_mfma)
                                     if t_hbm > t_mfma else "mfma: %.1f ms against %.1f ms for the measured traffic at 6.3 TB/s" % (t_mfma, t_hbm))
    if node:
        out["tree_nodes"] = {"launches": len(node), "avg_ms": round(float(np.mean(node)), 4),
                             "what": "npw_dtpqrt_batched, per node (two stacked 4096^2 triangles)"}
    return out


def first_contact(comm, args):
    """N > 1 (or the forced distributed path): the all-pairs link calibration of dist.link_calibration, before the first timed
    step and outside every timed region.  The transfer size follows the tile (128 MiB for the 4096^2 fp64 tile)."""
    from numpywren_amd import dist
    nbytes = min(128 << 20, max(1 << 20, args.tile * args.tile * 8))
    try:
        return dist.link_calibration(comm, nbytes=nbytes)
    except Exception as exc:
        print(f"[bench] rank {comm.rank}: link calibration failed: {exc!r}", file=sys.stderr, flush=True)
        watch = getattr(comm, "stall_watch", None)
        if "aborted" in repr(exc):
            # the communicator is gone: a timed run on it would only hang where the calibration did
            raise SystemExit(f"bench.py --gpus {comm.world}: the first payload over this node's links did not complete -- see the "
                             f"stall report of every rank above")
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["chol", "tsqr", "gemm32"], default="chol",
                    help="chol = the headline (BASELINE.json configs[1] / [2]); tsqr = configs[3]; gemm32 = configs[4]")
    ap.add_argument("--tiles", type=int, default=0, help="tiles per side (default: 4 on one GPU, 16 on several)")
    ap.add_argument("--leaves", type=int, default=0, help="tsqr: number of 4096-row leaves (default 256: configs[3]'s 1048576 x 4096 matrix, on any number of GPUs)")
    ap.add_argument("--tile", type=int, default=TILE)
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams per rank (0 = 1 on one GPU, 3 with several: a task waiting for a tile in "
                         "transit must not block the tasks behind it)")
    ap.add_argument("--priority-stream", action="store_true", help="panel kernels on a high-priority stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--r-only", action="store_true", help="tsqr: the timed run drops the V / T factors (no task reads them) instead of producing them")
    ap.add_argument("--keep-vt", action="store_true", help="tsqr: (the default since round 4; kept for old command lines)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 65536^2 single-GPU run of the N = 1 line")
    ap.add_argument("--no-anchor", action="store_true", help="N > 1: skip rank 0's one-GPU run of the same problem (config.one_gpu_anchor)")
    ap.add_argument("--dry-run", action="store_true", help="start / join the ranks, print who joined as one JSON line, do no GPU work")
    args = ap.parse_args()
    # dmabuf IPC between the ranks' processes (what RCCL needs on this driver stack); read by the HSA runtime when it starts, so it
    # has to be in the environment before the first HIP call of this process -- a launcher normally exports it, this is the net
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])     # does not return: this process becomes the launcher of N ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        # never a line whose n_gpus differs from the number of ranks in the job
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a job of WORLD_SIZE={world}: launch it as "
                         f"`python bench.py --gpus {args.gpus}` (it starts its own ranks) or under "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus}`")
    if args.dry_run:
        from numpywren_amd import dist
        comm = dist.init_process_group() if (world > 1 or os.environ.get("NUMPYWREN_AMD_FORCE_DIST")) else None
        joined = job_identity(comm, world, rank)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": args.gpus, "ranks_joined": len([j for j in joined if j]),
                              "transport": comm.backend if comm is not None else "none", "ranks": joined}))
        if comm is not None:
            comm.shutdown()
        return
    os.environ.pop("NUMPYWREN_AMD_STORE", None)
    if args.streams <= 0:
        # (tsqr: one stream -- a batch of 32 factorisations fills the chip by itself; two batches side by side only fight for
        #  workgroup slots: 1716 ms with one stream, 1940 with two on the round-3 build, gpurun_out/r03i.  gemm32: two -- the
        #  add_matrices tree of the reference's program is HBM-bound and its kernels need a dozen registers, so they run BESIDE
        #  the products of the other stream instead of between them: round 6, same box, 128.2 / 131.3 TFLOP/s with one stream,
        #  136.8 / 136.1 with two, 136.4 / 135.7 with three)
        args.streams = (2 if args.workload == "gemm32" else 1) if world == 1 else 3

    from numpywren_amd import alg_wrappers
    from numpywren_amd.device import get_backend
    from numpywren_amd.matrix import BigMatrix

    comm = None
    if world > 1 or os.environ.get("NUMPYWREN_AMD_FORCE_DIST"):   # the env var exercises the N > 1 code path on 1 GPU
        from numpywren_amd import dist
        # control: gloo; payload: RCCL over xGMI (npw_comm_*), one process per GPU.  The communicator itself is only made
        # after rank 0's one-GPU anchor run (below): a live communicator makes the library leave compute units to its
        # transfer kernels, which the anchor -- the plain one-GPU executor with its chain partition -- must not pay for
        comm = dist.init_process_group(open_transport=False)
    be = get_backend()
    b = args.tile
    from numpywren_amd import config as npw_config
    chain_cus = int(npw_config.default()["executor"].get("chain_cus", 0) or 0) if (world == 1 and args.streams == 1) else 0
    run = Runner(be, comm, args.streams, args.priority_stream)
    anchor = None
    calib = None     # N > 1: dist.link_calibration's result (first_contact), measured before the first timed step
    par = "1 gpu" if world == 1 else (f"{world} gpus, one process each, tiles 2-D block-cyclic, " + (
        "RCCL p2p panel exchange (npw_comm_*)" if comm is not None and comm.backend == "rccl" else
        "payloads staged through the host over the control group (ranks share a GPU: not a production transport)"))

    if args.workload == "chol":
        # N = 1: configs[1] (16384^2).  N > 1: STRONG scaling of configs[2]'s 65536^2 matrix (it fits one GPU: 34 GB).
        nb = args.tiles or (4 if world == 1 else 16)
        n = nb * b
        owner = comm.owner_fn(nb) if comm is not None else None
        if world > 1 and not args.no_anchor:
            # The anchor of THIS line's strong-scaling point: the same matrix on ONE GPU of this box (rank 0, the plain
            # one-GPU executor; the other ranks wait at the barrier) -- the N = 1 bench line's `value` is configs[1]'s
            # 16384^2 matrix, another problem, so a curve must be built from these anchors (or from that line's
            # `north_star`), never from its `value`.
            anchor = one_gpu_anchor(be, comm, rank, f"the same {n}x{n} matrix",
                                    lambda: (build_input(be, nb, b, f"bench_chol_anchor_{n}_{b}"),),
                                    lambda Xa: alg_wrappers.cholesky(Xa), n ** 3 / 3.0)
        if comm is not None:
            comm.open_transport()
            calib = first_contact(comm, args)
        X = build_input(be, nb, b, f"bench_chol_{n}_{b}", rank, world, owner)
        # inside the timed region only the roofline kernel is bracketed with events (an event record costs ~4 us of
        # stream time); the other kinds are timed in two extra steps after it, for `kernel_ms`
        elapsed, meta = run.timed(lambda: alg_wrappers.cholesky(X), args.steps, args.warmup, timers=("syrk",))
        run_step_ms = run.step_ms
        flops = n ** 3 / 3.0
        value = args.steps * flops / elapsed / 1e12
        line = {"metric": "achieved fp64 TFLOP/s, N x N tiled Cholesky (N^3/3 / wall)", "value": round(value, 3),
                "unit": "TFLOP/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"{n}x{n} fp64 Cholesky, {b}^2 tiles, {nb}x{nb} tile grid, alg_wrappers.cholesky "
                                       f"via LambdaPACK DAG ({nb*(nb+1)*(nb+2)//6} tasks)",
                           "n": n, "tile": b, "streams": args.streams, "chain_cus": chain_cus, "parallelism": par,
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
                                    # HBM-side bytes per launch: PMC passes of ANOTHER run of this command (2 x FETCH_SIZE +
                                    # WRITE_SIZE, profiles/), not measured in this process; only for the 4096^2 tile
                                    "traffic": (round(_syrk_traffic()[0]) if _syrk_traffic()[0] else None) if b == TILE else None,
                                    "traffic_unit": "B per tile update; from %s (rocprofv3 --pmc, separate passes of this command), "
                                                    "not this run (algorithmic 5.37e8)" % _syrk_traffic()[1],
                                    "launches": len(syrk), "avg_ms": round(avg_ms, 4),
                                    "algorithmic_flop_per_launch": 2 * b ** 3}
                # launches of the same kernel on the 192-CU partition beside a chol on the other 64 (kernel_ms
                # "syrk@rest" / "chol@chain") are not full-chip launches and stay out of the roofline average
                part = times.get("syrk@rest", [])
                if part:
                    line["roofline"]["beside_chol"] = {"launches": len(part), "avg_ms": round(float(np.mean(part)), 4),
                                                       "cus": be.compute_units - chain_cus}
                extra = 2
                run.timed(lambda: alg_wrappers.cholesky(X), extra, 0,
                          timers=("syrk", "syrk_sym", "trsm", "trsm_batch", "chol", "trtri_complete", "is_zero"))
                times = be.collect_kernel_times()
                line["kernel_ms"] = {k: round(float(np.mean(v)), 4) for k, v in times.items() if v}
                # what the timed kinds add up to per step (a window counts once, by its longer side)
                per_step = {k: float(np.sum(v)) / extra for k, v in times.items() if v}
                window = max(per_step.get("chol@chain", 0.0), per_step.get("syrk@rest", 0.0) + per_step.get("syrk_sym@rest", 0.0))
                busy = sum(v for k, v in per_step.items() if "@" not in k) + window
                line["kernel_ms"]["sum_per_step"] = round(busy, 3)
            # parity guard at full size: || A - L L^T ||_F / || A ||_F over ALL tiles (device side)
            line["config"]["residual_all_tiles"] = cholesky_residual(be, X, meta["outputs"][0], nb, full=True)
            if not args.no_north_star and b == TILE and nb == 4:
                # the north-star configuration on one GPU: configs[2]'s 65536^2 matrix (816 tasks), a few steps
                X.free()
                for m in meta["outputs"] + meta["intermediates"]:
                    m.free()
                nb2 = 16
                n2 = nb2 * b
                X2 = build_input(be, nb2, b, f"bench_chol_{n2}_{b}")
                ns_steps = 3
                e2, meta2 = run.timed(lambda: alg_wrappers.cholesky(X2), ns_steps, 1, timers=("syrk",))
                t2 = be.collect_kernel_times().get("syrk", [])
                tf = ns_steps * (n2 ** 3 / 3.0) / e2 / 1e12
                line["north_star"] = {"n": n2, "tile": b, "tasks": nb2 * (nb2 + 1) * (nb2 + 2) // 6, "steps": ns_steps,
                                      "ms_per_step": round(e2 / ns_steps * 1e3, 2), "tflops": round(tf, 3),
                                      "pct_peak": round(100 * tf / FP64_MFMA_PEAK_TFLOPS, 2),
                                      "syrk_frac": round(2.0 * b ** 3 / (float(np.mean(t2)) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4) if t2 else None,
                                      "syrk_launches": len(t2),
                                      "residual_all_tiles": cholesky_residual(be, X2, meta2["outputs"][0], nb2, full=True)}
                line["north_star"].update(step_stats(run.step_ms, line["north_star"]["ms_per_step"], "north_star.ms_per_step"))
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(b)
    elif args.workload == "tsqr":
        # configs[3]: (leaves * 4096) x 4096 fp64 TSQR; leaves in contiguous chunks per GPU, log2(world) exchanged R factors.
        # Every N runs the same problem (strong scaling) and returns what the reference's wrapper returns, [R, V, T]
        # (alg_wrappers.py:47): `value` is that run.  V and T of the 511 nodes are 160 GiB beside 32 GiB of input and 64 GiB
        # of R factors; since round 4 (one scratch buffer per stream instead of one per call) that fits one 288 GB GPU
        # without touching the host tier (1.73 s; 7.9 s through it in round 3).  On one GPU the line carries the R-ONLY run
        # (executor.drop_unread_outputs: V / T, which no task reads, are neither assembled nor stored) beside it, as
        # `config.r_only`, never instead of it.  --r-only makes it the timed run (and says so in the workload string).
        leaves = args.leaves or 256
        m = leaves * b
        r_only = args.r_only
        run.r_only = r_only
        if world > 1 and not args.no_anchor:
            def all_leaves():
                Xa = BigMatrix(f"bench_tsqr_anchor_{m}", shape=(m, b), shard_sizes=(b, b))
                for j in range(leaves):
                    Xa.put_tile(be.fill_random((b, b), 7, j * b, 0), j, 0)
                be.synchronize()
                return (Xa,)
            anchor = one_gpu_anchor(be, comm, rank, f"the same {m}x{b} TSQR" + (" (R only)" if r_only else " (R, V, T)"), all_leaves,
                                    lambda Xa: alg_wrappers.tsqr(Xa), 2.0 * m * b * b - 2.0 * b ** 3 / 3, r_only=r_only)
        if comm is not None:
            comm.open_transport()
            calib = first_contact(comm, args)
        if comm is not None:
            from numpywren_amd import dist
            comm.ownership = dist.tsqr_ownership(world, leaves)
        X = BigMatrix(f"bench_tsqr_{m}", shape=(m, b), shard_sizes=(b, b))
        for j in range(leaves):
            if comm is None or comm.owner("A", (j, 0)) == rank:
                X.put_tile(be.fill_random((b, b), 7, j * b, 0), j, 0)
        be.synchronize()
        elapsed, meta = run.timed(lambda: alg_wrappers.tsqr(X), args.steps, args.warmup, timers=("geqrt_batched", "tpqrt_batched"))
        qr_times = be.collect_kernel_times() if comm is None else {}
        run_step_ms = run.step_ms
        flops = 2.0 * m * b * b - 2.0 * b ** 3 / 3
        value = args.steps * flops / elapsed / 1e12
        line = {"metric": "achieved fp64 TFLOP/s, m x n TSQR ((2 m n^2 - 2 n^3 / 3) / wall)", "value": round(value, 3),
                "unit": "TFLOP/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"{m}x{b} fp64 TSQR, {leaves} leaves, alg_wrappers.tsqr ({2 * leaves - 1} tasks), "
                                       + ("R only (V / T dropped on store)" if r_only else "R, V, T kept (the reference's outputs)"),
                           "r_only": r_only, "tile": b, "streams": args.streams, "parallelism": par}}
        if comm is None:
            line["roofline"] = tsqr_roofline(qr_times, b, r_only)
        if comm is None and not r_only:
            run.r_only = True
            e2, _ = run.timed(lambda: alg_wrappers.tsqr(X), args.steps, 1, timers=("geqrt_batched", "tpqrt_batched"))
            t2 = be.collect_kernel_times()
            run.r_only = False
            line["config"]["r_only_run"] = {"what": "the same program with executor.drop_unread_outputs: only the R factors are produced",
                                            "ms_per_step": round(e2 / args.steps * 1e3, 3),
                                            "tflops": round(args.steps * flops / e2 / 1e12, 3)}
            line["config"]["r_only_run"].update(step_stats(run.step_ms, line["config"]["r_only_run"]["ms_per_step"], "r_only_run.ms_per_step"))
            line["config"]["r_only_run"]["roofline"] = tsqr_roofline(t2, b, True)
        if comm is None and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_tsqr(b, leaves)
    else:
        # configs[4]: 32768^2 fp32 GEMM program (fp32 MFMA products, the reference's fp64 add_matrices tree), strong scaling
        nb = args.tiles or 8
        n = nb * b
        if world > 1 and not args.no_anchor:
            def all_tiles():
                Aa = BigMatrix(f"bench_gA_anchor_{n}", shape=(n, n), shard_sizes=(b, b), dtype=np.float32)
                Ba = BigMatrix(f"bench_gB_anchor_{n}", shape=(n, n), shard_sizes=(b, b), dtype=np.float32)
                for i in range(nb):
                    for j in range(nb):
                        Aa.put_tile(be.convert(be.fill_random((b, b), 11, i * b, j * b), np.float32), i, j)
                        Ba.put_tile(be.convert(be.fill_random((b, b), 12, i * b, j * b), np.float32), i, j)
                be.synchronize()
                return (Aa, Ba)
            anchor = one_gpu_anchor(be, comm, rank, f"the same {n}x{n} fp32 GEMM program", all_tiles,
                                    lambda Aa, Ba: alg_wrappers.gemm(Aa, Ba), 2.0 * n ** 3)
        if comm is not None:
            from numpywren_amd import dist
            comm.open_transport()
            calib = first_contact(comm, args)
            comm.ownership = dist.gemm_ownership(world)
        A = BigMatrix(f"bench_gA_{n}", shape=(n, n), shard_sizes=(b, b), dtype=np.float32)
        B = BigMatrix(f"bench_gB_{n}", shape=(n, n), shard_sizes=(b, b), dtype=np.float32)
        for i in range(nb):
            for j in range(nb):
                if comm is None or comm.owner("A", (i, j)) == rank:
                    A.put_tile(be.convert(be.fill_random((b, b), 11, i * b, j * b), np.float32), i, j)
                if comm is None or comm.owner("B", (i, j)) == rank:
                    B.put_tile(be.convert(be.fill_random((b, b), 12, i * b, j * b), np.float32), i, j)
        be.synchronize()
        elapsed, meta = run.timed(lambda: alg_wrappers.gemm(A, B), args.steps, args.warmup, timers=("gemm",))
        gemm_iv = be.collect_kernel_times(intervals=True).get("gemm", []) if comm is None else []
        gemm_times = [(e - s0) / max(1, c) for s0, e, c in gemm_iv]      # per product: a batched launch of c products / c
        gemm_products = sum(max(1, c) for _, _, c in gemm_iv)
        run_step_ms = run.step_ms
        value = args.steps * 2.0 * n ** 3 / elapsed / 1e12
        line = {"metric": "achieved fp32 TFLOP/s, N x N GEMM program (2 N^3 / wall)", "value": round(value, 3),
                "unit": "TFLOP/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{n}x{n} fp32 GEMM, {b}^2 tiles, alg_wrappers.gemm (fp32 MFMA products, fp64 "
                                       f"add_matrices tree as in the reference)", "tile": b, "streams": args.streams,
                           "parallelism": par, "pct_fp32_mfma_peak": round(100 * value / (157.3 * args.gpus), 2)}}
        if comm is None and gemm_times:
            avg_ms = float(np.mean(gemm_times))
            # With two executor streams (the default for this workload: the HBM-bound add_matrices tree then runs beside the
            # products instead of between them) two products share the chip and each takes twice as long: the kernel's rate is
            # the flops of all launches over the time at least one of them was running, and `concurrency` says how many were.
            busy, cur_s, cur_e = 0.0, None, None
            for s0, e, _ in sorted(gemm_iv):
                if cur_e is None or s0 > cur_e:
                    busy += (cur_e - cur_s) if cur_e is not None else 0.0
                    cur_s, cur_e = s0, e
                else:
                    cur_e = max(cur_e, e)
            busy += (cur_e - cur_s) if cur_e is not None else 0.0
            concurrency = sum(e - s0 for s0, e, _ in gemm_iv) / busy if busy > 0 else 1.0
            achieved = gemm_products * 2.0 * b ** 3 / (busy * 1e-3) / 1e12
            pmc = _profile_json("r06_gemm32_pmc.json") or {}
            line["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel<float,128,128,32,true,true,false,0,2> (kernels.gemm on two fp32 tiles: "
                                                           "one 4096^3 product = 1024 workgroups; the executor hands the ready products over in "
                                                           "launches of up to 16, npw_sgemm_batched; B tiles transposed once, then the N / T form)",
                                "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "launches": len(gemm_iv), "products": gemm_products, "avg_ms": round(avg_ms, 4),
                                "avg_ms_is": "per 4096^3 product (a launch's duration / the products it holds)",
                                "concurrency": round(concurrency, 3),
                                "achieved_is": "flops of all launches / time at least one was running (HIP events on the launching streams, one clock); "
                                               "avg_ms is a launch's own duration -- with `concurrency` launches sharing the chip",
                                "algorithmic_flop_per_launch": 2 * b ** 3,
                                "traffic": pmc.get("bytes_per_launch"),
                                "traffic_unit": "B per 4096^3 product; from profiles/r06_gemm32_pmc.json (rocprofv3 --pmc, separate passes of "
                                                "this command), not this run (algorithmic 3 x 64 MiB = 2.01e8)"}
        if comm is None and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_gemm32(b, nb)
        if comm is None:
            # beside the parity mode (`value`): the same program with executor.fuse_gemm_reduction -- the K partial
            # products of a C tile accumulate in one fp32 buffer, no Temp tiles, no add_matrices tree
            run.fuse = True
            e2, _ = run.timed(lambda: alg_wrappers.gemm(A, B), args.steps, 1)
            run.fuse = False
            fused = args.steps * 2.0 * n ** 3 / e2 / 1e12
            line["config"]["fused_tflops"] = round(fused, 3)
            line["config"]["fused_ms_per_step"] = round(e2 / args.steps * 1e3, 3)
            line["config"]["fused_step_ms"] = run.step_ms
            line["config"]["fused_pct_fp32_mfma_peak"] = round(100 * fused / 157.3, 2)
            line["config"]["fused_mode"] = "executor.fuse_gemm_reduction: fp32 accumulation in place, result converted to fp64 once"
    line.update(step_stats(run_step_ms, line["ms_per_step"]))
    if comm is not None:
        line["config"]["transport"] = comm.backend
        if calib is not None:
            # measured before the first timed step, outside the timed region: what one 128 MiB tile costs on this node's links
            line["config"]["xgmi_GBps"] = calib["xgmi_GBps"]
            line["config"]["link_calibration"] = {k: v for k, v in calib.items() if k not in ("xgmi_GBps", "samples")}
            line["config"]["link_calibration"]["samples"] = calib["samples"][:64]
            if world > 1 and rank == 0:
                line["config"]["predicted_at_measured_link"] = _predicted_at_link(args.workload, world, calib["xgmi_GBps"]["median"])
        # who ran this: the ranks that joined the control group, the device each one bound, and the size RCCL itself
        # reports for the payload communicator -- a line is only printed when they all equal --gpus
        joined = job_identity(comm, world, rank)
        line["config"]["ranks_joined"] = len([j for j in joined if j])
        line["config"]["devices"] = [j and j["device"] for j in joined]
        line["config"]["rccl_nranks"] = joined[0]["rccl_nranks"] if joined and joined[0] else None
        if line["config"]["ranks_joined"] != args.gpus or (comm.backend == "rccl" and (
                line["config"]["rccl_nranks"] != args.gpus or len(set(line["config"]["devices"])) != args.gpus)):
            raise SystemExit(f"bench.py --gpus {args.gpus}: {line['config']['ranks_joined']} ranks joined, RCCL reports "
                             f"{line['config']['rccl_nranks']}, devices {line['config']['devices']} -- not printing a line "
                             f"labelled {args.gpus} GPUs")
        if anchor is not None:
            line["config"]["one_gpu_anchor"] = anchor
        # The line explains itself (no multi-GPU node was ever available to the builder; the first real run is the driver's):
        # per rank, what the last TIMED step's common walk cost on the host (`host_walk_ms`), how long the host was blocked on
        # the device or the control group, how far the device ran behind it (`drain_ms`), and the bytes moved; then ONE extra
        # step outside the timed region with executor.task_timers: device time by kernel name (`kernel_busy_ms`) and the
        # transport stream's time inside exchanges (`transfer_wait_ms`: waiting for producers and peers included).
        timed_diag = (run.last_dist or {}).get("diag")
        run.task_timers = True
        build = {"chol": (lambda: alg_wrappers.cholesky(X)), "tsqr": (lambda: alg_wrappers.tsqr(X)),
                 "gemm32": (lambda: alg_wrappers.gemm(A, B))}[args.workload]
        e_diag, _ = run.timed(build, 1, 0)
        run.task_timers = False
        extra_diag = dict((run.last_dist or {}).get("diag") or {}, step_ms=round(e_diag * 1e3, 3))
        gathered = [None] * world
        if world > 1:
            comm.dist.all_gather_object(gathered, {"timed_step": timed_diag, "diagnostic_step": extra_diag})
        else:
            gathered = [{"timed_step": timed_diag, "diagnostic_step": extra_diag}]
        line["per_rank"] = gathered
        predicted = _predicted_scaling(args.workload)
        if predicted is not None:
            # the N = 1 point of the Cholesky curve is not the N = 1 line's `value` (configs[1]: 16384^2, what BASELINE.json
            # asks that line to report) but its `north_star` object: the same 65536^2 matrix on one GPU
            line["config"]["predicted"] = predicted
        if world > 1 and comm.backend != "rccl" and not os.environ.get("NUMPYWREN_AMD_DIST_BACKEND"):
            raise SystemExit("bench.py --gpus %d: the ranks ended up on the host-staged transport (%s) -- tiles would cross "
                             "PCIe and host memory instead of xGMI; a number from this run would mean nothing.  Set "
                             "NUMPYWREN_AMD_DIST_BACKEND=gloo to force it knowingly.  Who joined (rank, LOCAL_RANK, devices this "
                             "process sees, the *_VISIBLE_DEVICES it was given, PCI bus id of the device it bound): %s"
                             % (world, comm.backend, [(j["rank"], j["local_rank"], j["visible_devices"], j["env"], j["device"])
                                                      for j in joined if j]))
    if rank == 0:
        print(json.dumps(line))
    if comm is not None:
        comm.shutdown()


if __name__ == "__main__":
    main()
