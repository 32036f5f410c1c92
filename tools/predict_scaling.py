#!/usr/bin/env python
"""Developer aid (CPU only): PREDICTED strong-scaling figures of the three multi-GPU workloads of BASELINE.json on 1 / 2 / 4 / 8
GPUs from MEASURED single-GPU kernel times -- something to hold the driver's first multi-GPU run against (no multi-GPU node
has been available to the builder).  Writes profiles/predicted_scaling.json: the ONE table bench.py's N > 1 lines, DESIGN.md
section 6 and BASELINE.md quote.      python tools/predict_scaling.py [--link-gbs 64] [--write]

configs[2], 65536^2 Cholesky (16 x 16 tiles of 4096^2, 816 tasks): a list-scheduling simulation of numpywren_amd/dist.py:
  * the common task sequence = LambdaPackProgram's ready heap (critical-path priority), children released when their
    parents have been issued; EVERY rank walks every position of it on the host and can only enqueue a task once its walk has
    reached it.  The walk's cost is MEASURED on the real backend (round 5, tools/dist_host_split.py on one MI355X,
    profiles/r05_dist_host_split.md): 20 us of bookkeeping per position of the common sequence on every rank (dequeue,
    look-ups, exchange plan, post_op) + 55 us per task the rank runs itself (ctypes marshalling, events, allocator) -- 61 ms
    for the world of one (816 + 816), 24 - 29 ms for rank 0 of 8 (816 + 100; the stand-in run that executes only that
    rank's tasks).  Round 4 charged 101 us per position to everybody, from 8 gloo ranks on the CHECKER backend;
  * tile ownership 2-D block-cyclic on the Pr x Pc grid, owner computes; a GPU runs one chip-filling kernel at a time and
    picks, among its tasks whose inputs have arrived, the earliest in the common sequence (3 executor streams);
  * a produced tile is pushed to every GPU owning a consumer: 128 MiB per destination, one xGMI link per pair of GPUs,
    transfers on one link serialised, different links in parallel, at `--link-gbs` per direction (default 64 GB/s: xGMI's
    153.6 GB/s per link is bidirectional, RCCL point-to-point reaches somewhat less than the 76.8 GB/s per direction);
  * kernel times: the N = 1 row uses what one in-order stream with the chain partition measures (bench.py's `kernel_ms`);
    the N > 1 rows the times of the 3-stream, no-chain-partition configuration those ranks run (`bench.py --streams 3`:
    a chol there is fenced -- it gets the chip -- but the batched launches are shorter because tasks become ready one by one).
configs[3], 1048576 x 4096 TSQR (256 leaves): per GPU the leaf batches and the local tree levels at the measured batched
  times (tools/qr_soak.py, tools/tpqrt_time.py, with T: bench.py keeps R, V, T on any number of GPUs), then log2(N) levels
  of one 128 MiB R factor over one link + one single-node factorisation each.
configs[4], 32768^2 fp32 GEMM program (8 x 8 x 8 tiles): C tiles 2-D block-cyclic, the prologue's A / B panel pushes as one
  all-to-all-v (bytes of the busiest directed link / link rate; products start as their operands arrive, so the run is the
  longer of the two plus the first tile's transfer), then 512 / N products + the add_matrices trees at the measured rates.
The models have no RCCL launch latency, no contention between transfer kernels and compute kernels, and no HBM effect of
concurrent receives: a first real run well below them points at one of those."""
import argparse
import heapq
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if __name__ == "__main__":
    os.environ["NUMPYWREN_AMD_STORE"] = "host"     # (not when bench.py imports the model: no tile is stored by it anyway)

# ---- measured single-GPU inputs (ms); sources in profiles/r04_*.md ----------------------------------------------------
KERNEL_MS_1GPU = {"chol": 1.48, "trsm": 1.10, "syrk": 1.89, "syrk_sym": 1.08}          # one stream + chain partition
KERNEL_MS_3STREAMS = {k: round(v * 1359.4 / 1343.5, 4) for k, v in KERNEL_MS_1GPU.items()}   # bench.py --tiles 16 --streams 3: 1359.4 ms against 1343.5 (gpurun_out/r04l): every kind scaled by that ratio
HOST_US_PER_POSITION = 20.0    # every rank, every position of the common sequence (tools/dist_host_split.py, real backend)
HOST_US_PER_OWN_TASK = 55.0    # ... plus this for a task the rank runs itself
TILE_BYTES = 4096 * 4096 * 8
# batched QR of 4096^2 tiles, ms per call by batch size: dense leaves / stacked-triangle tree nodes, with T and R only
GEQRT_MS = {True: {1: 18.4, 2: 21.5, 4: 27.1, 8: 36.4, 16: 60.0, 32: 105.0}, False: {1: 16.5, 2: 19.5, 4: 24.0, 8: 31.0, 16: 47.2, 32: 77.8}}
TPQRT_MS = {True: {1: 18.8, 2: 20.5, 4: 25.6, 8: 35.1, 16: 52.7, 32: 87.9}, False: {1: 17.5, 2: 19.0, 4: 22.5, 8: 27.0, 16: 36.2, 32: 56.9}}
SGEMM_TILE_MS = 2 * 4096 ** 3 / 136.3e12 * 1e3      # 32768^2 fp32 program at 136.3 TFLOP/s in the parity mode, adds included (round 6: two
                                                    # executor streams, the add tree beside the products; 131.5 with one stream until round 5)


def batched_ms(table, count):
    """time of `count` independent factorisations issued as batches of at most 32 (a batch between two measured sizes is
    charged as the next larger one)"""
    total, left = 0.0, count
    while left > 0:
        take = min(32, left)
        total += table[min(k for k in table if k >= take)]
        left -= take
    return total


def simulate_cholesky(world, nb, link_gbs, kernel_ms, host_us, host_own_us=HOST_US_PER_OWN_TASK):
    from numpywren_amd import alg_wrappers
    from numpywren_amd.dist import process_grid
    from numpywren_amd.matrix import BigMatrix
    X = BigMatrix(f"sim_{world}_{nb}", shape=(nb * 8, nb * 8), shard_sizes=(8, 8))
    program, meta = alg_wrappers.cholesky(X)
    compiled = program.program
    tasks = compiled.tasks
    pr, pc = process_grid(world)

    def owner(name, idx):
        return (idx[-2] % pr) * pc + (idx[-1] % pc)

    prio = program._priorities()
    name_of = lambda t: getattr(compiled.kernel(t.expr_idx), "__name__", "")

    def cost(t):
        k = name_of(t)
        if k == "syrk" and t.reads[1] == t.reads[2]:
            k = "syrk_sym"
        return kernel_ms[k]

    # common sequence
    nparents = {t.index: len({p.index for p in t.parents}) for t in tasks}
    ready = [(-prio[t.key], t.index) for t in tasks if nparents[t.index] == 0]
    heapq.heapify(ready)
    seq = []
    while ready:
        _, i = heapq.heappop(ready)
        t = tasks[i]
        seq.append(t)
        for c in {c.index: c for c in t.children}.values():
            nparents[c.index] -= 1
            if nparents[c.index] == 0:
                heapq.heappush(ready, (-prio[c.key], c.index))
    assert len(seq) == len(tasks)
    pos = {t.index: n for n, t in enumerate(seq)}
    rank_of = {t.index: owner(*t.writes[0]) for t in tasks}
    # when rank r's walk has passed position p: bookkeeping for every position, the enqueue work for its own tasks
    reach = [[0.0] * len(seq) for _ in range(world)]
    for r in range(world):
        clock = 0.0
        for n_, t in enumerate(seq):
            clock += host_us * 1e-3 + (host_own_us * 1e-3 if rank_of[t.index] == r else 0.0)
            reach[r][n_] = clock
    xfer_ms = TILE_BYTES / (link_gbs * 1e9) * 1e3
    finish = {}                       # task index -> finish time
    arrive = {}                       # (tile, rank) -> arrival time
    link_free = {}                    # (src, dst) -> time
    gpu_free = [0.0] * world
    pending = {r: [] for r in range(world)}
    for t in seq:
        pending[rank_of[t.index]].append(t)
    done = set()
    sent_bytes = 0
    remaining = len(tasks)
    while remaining:
        progressed = False
        for r in range(world):
            best = None
            for t in pending[r][:64]:      # (a GPU runs ahead of the common sequence by a bounded window)
                ok, when = True, reach[r][pos[t.index]]           # the host's walk has to have reached the task
                for rd in t.reads:
                    w = compiled.writer_of(*rd)
                    if w is None:
                        continue           # an input tile: resident (the prologue moves none for this program)
                    if w.index not in done:
                        ok = False
                        break
                    when = max(when, arrive[(rd, r)])
                if ok and (best is None or max(when, gpu_free[r]) < best[0] - 1e-12):
                    best = (max(when, gpu_free[r]), t)
                    if when <= gpu_free[r]:
                        break              # nothing can start earlier than "now"
            if best is None:
                continue
            start, t = best
            end = start + cost(t)
            gpu_free[r] = end
            finish[t.index] = end
            done.add(t.index)
            pending[r].remove(t)
            remaining -= 1
            progressed = True
            for wtile in t.writes:
                arrive[(wtile, r)] = end
                dests = sorted({rank_of[c.index] for c in t.children if wtile in c.reads} - {r})
                for d in dests:
                    s = max(end, link_free.get((r, d), 0.0))
                    link_free[(r, d)] = s + xfer_ms
                    arrive[(wtile, d)] = s + xfer_ms
                    sent_bytes += TILE_BYTES
        if not progressed:
            raise RuntimeError("simulation stalled")
    total = max(finish.values())
    n = nb * 4096
    busy = sum(cost(t) for t in tasks)
    return {"gpus": world, "grid": f"{pr}x{pc}", "ms": round(total, 1), "tflops": round(n ** 3 / 3 / (total * 1e-3) / 1e12, 1),
            "host_walk_ms": round(max(reach[r][-1] for r in range(world)), 1), "sum_of_kernel_ms": round(busy, 1),
            "efficiency_vs_sum": round(busy / world / total, 3), "GB_moved": round(sent_bytes / 1e9, 1)}


def predict_tsqr(world, leaves, link_gbs):
    keep_vt = True                             # bench.py: R, V, T -- the reference's outputs -- on any number of GPUs
    per = leaves // world
    ms = batched_ms(GEQRT_MS[keep_vt], per)
    nodes = per // 2
    while nodes >= 1:                          # the local tree: levels of per/2, per/4, ... 1 nodes
        ms += batched_ms(TPQRT_MS[keep_vt], nodes)
        nodes //= 2
    xfer = TILE_BYTES / (link_gbs * 1e9) * 1e3
    cross = int(math.log2(world)) if world > 1 else 0
    ms += cross * (xfer + TPQRT_MS[keep_vt][1])
    m, n = leaves * 4096, 4096
    flops = 2.0 * m * n * n - 2.0 * n ** 3 / 3
    return {"gpus": world, "ms": round(ms, 1), "tflops": round(flops / (ms * 1e-3) / 1e12, 1),
            "outputs": "R, V, T" if keep_vt else "R only", "cross_gpu_levels": cross,
            "GB_moved": round(cross * (world // 2 if world > 1 else 0) * TILE_BYTES / 1e9, 2)}


def predict_gemm(world, nb, link_gbs):
    from numpywren_amd.dist import process_grid
    pr, pc = process_grid(world)
    tile = 4096 * 4096 * 4
    # A[i, k] goes to the pc GPUs of grid row i, B[k, j] to the pr GPUs of grid column j; input tiles start block-cyclic
    link = {}
    for i in range(nb):
        for k in range(nb):
            home_a = (i % pr) * pc + (k % pc)
            for c in range(pc):
                d = (i % pr) * pc + c
                if d != home_a:
                    link[(home_a, d)] = link.get((home_a, d), 0) + tile
            home_b = (i % pr) * pc + (k % pc)          # B[i, k] (row index i is the contraction index here)
            for r in range(pr):
                d = r * pc + (k % pc)
                if d != home_b:
                    link[(home_b, d)] = link.get((home_b, d), 0) + tile
    prologue = (max(link.values()) / (link_gbs * 1e9) * 1e3) if link else 0.0
    first = tile / (link_gbs * 1e9) * 1e3 if link else 0.0
    compute = nb ** 3 / world * SGEMM_TILE_MS
    ms = max(compute, prologue) + first
    n = nb * 4096
    return {"gpus": world, "grid": f"{pr}x{pc}", "ms": round(ms, 1), "tflops": round(2.0 * n ** 3 / (ms * 1e-3) / 1e12, 1),
            "prologue_ms": round(prologue, 1), "compute_ms": round(compute, 1), "GB_moved": round(sum(link.values()) / 1e9, 1)}


def predict(workload, world, link_gbs):
    """One row of the table for `workload` on `world` GPUs at `link_gbs` GB/s per direction (bench.py re-evaluates the model with
    the link rate it measured on the node it runs on)."""
    if workload == "chol":
        return simulate_cholesky(world, 16, link_gbs, KERNEL_MS_1GPU if world == 1 else KERNEL_MS_3STREAMS, HOST_US_PER_POSITION)
    if workload == "tsqr":
        return predict_tsqr(world, 256, link_gbs)
    return predict_gemm(world, 8, link_gbs)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--link-gbs", type=float, default=64.0)
    ap.add_argument("--host-us", type=float, default=HOST_US_PER_POSITION, help="host bookkeeping per position of the common sequence, every rank")
    ap.add_argument("--host-own-us", type=float, default=HOST_US_PER_OWN_TASK, help="host work per task a rank runs itself")
    ap.add_argument("--tiles", type=int, default=16)
    ap.add_argument("--write", action="store_true", help="write profiles/predicted_scaling.json")
    a = ap.parse_args()
    out = {"note": "PREDICTED from measured single-GPU kernel times by tools/predict_scaling.py; never measured on more than one GPU",
           "link_GBps_per_direction": a.link_gbs, "host_us_per_position": a.host_us, "host_us_per_own_task": a.host_own_us,
           "host_model_source": "tools/dist_host_split.py on one MI355X, real backend (profiles/r05_dist_host_split.md): world of one "
                                "61 ms measured / 61.2 modelled, rank 0 of 8 24 - 29 ms measured / 21.8 + 3 (prologue) modelled",
           "workloads": {}}
    rows = []
    for w in (1, 2, 4, 8):
        rows.append(simulate_cholesky(w, a.tiles, a.link_gbs, KERNEL_MS_1GPU if w == 1 else KERNEL_MS_3STREAMS, a.host_us, a.host_own_us))
        print("chol  ", rows[-1])
    out["workloads"]["chol"] = {"what": "65536^2 fp64 Cholesky, 4096^2 tiles (bench.py --gpus N)",
                                "tflops_by_gpus": {str(r["gpus"]): r["tflops"] for r in rows}, "rows": rows,
                                "kernel_ms": {"1": KERNEL_MS_1GPU, "N>1 (3 streams, no chain partition)": KERNEL_MS_3STREAMS}}
    rows = [predict_tsqr(w, 256, a.link_gbs) for w in (1, 2, 4, 8)]
    for r in rows:
        print("tsqr  ", r)
    out["workloads"]["tsqr"] = {"what": "1048576 x 4096 fp64 TSQR, 256 leaves (bench.py --workload tsqr --gpus N)",
                                "tflops_by_gpus": {str(r["gpus"]): r["tflops"] for r in rows}, "rows": rows}
    rows = [predict_gemm(w, 8, a.link_gbs) for w in (1, 2, 4, 8)]
    for r in rows:
        print("gemm32", r)
    out["workloads"]["gemm32"] = {"what": "32768^2 fp32 GEMM program, 4096^2 tiles (bench.py --workload gemm32 --gpus N)",
                                  "tflops_by_gpus": {str(r["gpus"]): r["tflops"] for r in rows}, "rows": rows}
    if a.write:
        with open(os.path.join(ROOT, "profiles", "predicted_scaling.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("wrote profiles/predicted_scaling.json")
