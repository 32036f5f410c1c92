#!/usr/bin/env python
"""Developer aid (CPU only): predicted strong-scaling curve of the 65536^2 Cholesky (16 x 16 tiles of 4096^2, 816 tasks)
on 1 / 2 / 4 / 8 GPUs from MEASURED single-GPU kernel times -- something to hold the driver's first multi-GPU run against
(no multi-GPU node has been available to the builder).  A list-scheduling simulation of numpywren_amd/dist.py:

  * the common task sequence = LambdaPackProgram's ready heap (critical-path priority), children released when their
    parents have been issued;
  * tile ownership 2-D block-cyclic on the Pr x Pc grid, owner computes; a GPU runs one chip-filling kernel at a time and
    picks, among its tasks whose inputs have arrived, the earliest in the common sequence (3 executor streams);
  * a produced tile is pushed to every GPU owning a consumer: 128 MiB per destination, one xGMI link per pair of GPUs,
    transfers on one link serialised, different links in parallel, at `--link-gbs` per direction (default 64 GB/s: xGMI's
    153.6 GB/s per link is bidirectional, RCCL point-to-point reaches somewhat less than the 76.8 GB/s per direction).

Kernel times (ms per task, profiles/r03_bench_line.json, batched launches): chol 1.48, trsm 1.10, syrk (x is not y) 1.89,
syrk (x is y) 1.08.      python tools/predict_scaling.py [--link-gbs 64]"""
import argparse
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NUMPYWREN_AMD_STORE"] = "host"

KERNEL_MS = {"chol": 1.48, "trsm": 1.10, "syrk": 1.89, "syrk_sym": 1.08}
TILE_BYTES = 4096 * 4096 * 8


def simulate(world, nb, link_gbs):
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
        return KERNEL_MS[k]

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
    # event loop: repeatedly let every GPU start the earliest-in-sequence task whose inputs' arrival times are known
    remaining = len(tasks)
    while remaining:
        progressed = False
        for r in range(world):
            best = None
            for t in pending[r][:64]:      # (a GPU runs ahead of the common sequence by a bounded window)
                ok, when = True, 0.0
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
            "sum_of_kernel_ms": round(busy, 1), "efficiency_vs_sum": round(busy / world / total, 3),
            "GB_moved": round(sent_bytes / 1e9, 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--link-gbs", type=float, default=64.0)
    ap.add_argument("--tiles", type=int, default=16)
    a = ap.parse_args()
    for w in (1, 2, 4, 8):
        print(simulate(w, a.tiles, a.link_gbs))
