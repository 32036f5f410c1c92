#!/usr/bin/env python
"""Developer aid (CPU only): what one position of dist.py's common task walk costs on the host -- the 16 x 16-tile Cholesky
(816 tasks) of bench.py --gpus 8 on 8 gloo ranks with the checker backend and 4 x 4-element tiles (tests/test_dist_gloo.py's
"grid16" scenario), so that the arithmetic is negligible and what is left is the walk: dequeue, ownership, exchange plan,
post_op, and the host-staged 128-byte "transfers".  tools/predict_scaling.py takes its host cost per position from here."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    import torch.multiprocessing as mp
    import test_dist_gloo as t
    world = 8
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(t._worker, args=(world, t._free_port(), "grid16", d), nprocs=world, join=True)
        rows = [json.load(open(os.path.join(d, f"diag_{r}.json"))) for r in range(world)]
    for r in rows:
        print({k: r[k] for k in ("rank", "positions", "tasks_run_here", "host_walk_ms", "host_blocked_ms", "transfer_wait_ms")})
    per = [(r["host_walk_ms"] - r["transfer_wait_ms"]) / 816 for r in rows]
    print("host cost per position of the common sequence (816 tasks): %.3f - %.3f ms, mean %.3f (transfers' host time taken out)"
          % (min(per), max(per), sum(per) / len(per)))
