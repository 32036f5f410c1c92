"""Configuration.  The reference loads an AWS-centric YAML (numpywren/config.py:39-50); here
`default()` returns a local dict describing the single-node MI355X executor.  Keys can be
overridden with the environment variables listed below or a YAML file named by
$NUMPYWREN_AMD_CONFIG_FILE.
"""
import os

DEFAULTS = {
    "executor": {
        "streams": 4,            # HIP streams in the worker pool (job_runner pipeline_width analogue)
        # Issue the latency-bound panel kernels (chol / trsm / qr_factor) on a separate high-priority
        # stream.  Off by default: measured on MI355X (profiles/r01_overlap_study.md) a resident
        # 1024-workgroup trailing update leaves no free slot (LDS-full CUs, no preemption), so every
        # small panel kernel queues ~1 ms behind it and the critical path gets slower, not faster.
        "priority_stream": False,
        "exact_zero_shortcircuit": True,  # reproduce the reference's allclose(x, 0) early-outs
        "reclaim_intermediates": False,   # free tiles of non-input / non-output matrices after their last reader
        # also drop, as soon as they are stored, tiles of such matrices that NO task reads (TSQR's V / T factors, which
        # the reference's wrapper returns next to R, alg_wrappers.py:47): an explicit "R only" request, never implied
        "drop_unread_outputs": False,
        # GEMM program: accumulate the K partial products of a C tile in ONE buffer (beta = 1 on the GEMM's accumulator:
        # fp32 for fp32 tiles) instead of materialising Temp[i, j, k, l] and summing them with the fp64 add_matrices
        # tree (job_runner.ReductionFusion).  Off = the reference's arithmetic, task by task (the parity mode).
        "fuse_gemm_reduction": False,
        # Ready tasks of one latency-bound kind (qr_factor: the TSQR leaves, the nodes of a tree level) that are
        # handed to the device as a single batched launch sequence; 1 = one task at a time.  32 = what the QR panel kernel
        # holds at once for 4096-row tiles (2 workgroups per CU x 256 CUs / 16 slabs); 128-leaf TSQR: 1046 ms with 16,
        # 949 ms with 32 (profiles/r02_qr_tsqr.md).
        "batch_tasks": 32,
        # Compute units set aside for the panel chain while trailing updates are ready: a `chol` task that becomes
        # ready while independent trailing updates are still queued runs on a stream masked to this many CUs, the
        # updates issued next run on a stream masked to the others, and the one in-order stream resumes when the
        # factorisation is done (job_runner.LambdaPackExecutor.run_chain).  64 = the 63 panel workgroups of a 4096^2
        # tile + 1; the other 192 CUs hold a 1024-tile syrk in exactly 3 rounds.  0 = off.  $NUMPYWREN_AMD_CHAIN_CUS.
        "chain_cus": 64,
        # The host-DRAM tier (residency.py) reads the static DAG: victims by farthest next read and copies back ahead of the
        # readers (False: round 1's LRU, restores on demand); tiles of the next `spill_prefetch_tasks` tasks are brought
        # back early; while the tier is at work (a byte budget is set, or tiles have been pushed out) the batches of the
        # THROUGHPUT kernels (syrk, trsm, gemm) are at most `spill_batch_tasks` tasks: a batched launch waits for the
        # copy-in of ALL its operands, so 16 trailing updates per launch make copies and kernels take turns (32768^2
        # Cholesky, 12 tiles of budget: 404 ms with batches of 32, 328 with 8; 24 tiles: 264 / 204 with 4).
        # One roctx range per task / batch ("<kernel>(<node>)") around the calls that enqueue its kernels (npw_range_push / _pop):
        # `rocprofv3 --marker-trace` shows tasks beside kernels.  Off: two library calls per task.
        "roctx_ranges": False,
        "spill_plan": True,
        "spill_prefetch_tasks": 2,
        "spill_batch_tasks": 8,
    },
    "store": {
        "tier": "hbm",           # "hbm" (device memory) or "host" (pinned/pageable host memory)
        "device_pool": True,
        # Byte budget for tiles stored in HBM (int or "200G"); beyond it the tiles whose next read is farthest in the task
        # sequence move to pinned host DRAM and come back ahead of that read (residency.py).  None: no budget -- tiles are only pushed out
        # when a device allocation fails.  $NUMPYWREN_AMD_HBM_BUDGET overrides.
        "hbm_budget_bytes": None,
    },
    "runtime": {"bucket": "hbm"},
}


def default():
    import copy
    cfg = copy.deepcopy(DEFAULTS)
    path = os.environ.get("NUMPYWREN_AMD_CONFIG_FILE")
    if path and os.path.exists(path):
        import yaml
        with open(path) as f:
            user = yaml.safe_load(f) or {}
        for section, values in user.items():
            cfg.setdefault(section, {}).update(values or {})
    tier = os.environ.get("NUMPYWREN_AMD_STORE")
    if tier:
        cfg["store"]["tier"] = tier
    chain = os.environ.get("NUMPYWREN_AMD_CHAIN_CUS")
    if chain is not None and chain != "":
        cfg["executor"]["chain_cus"] = int(chain)
    streams = os.environ.get("NUMPYWREN_AMD_STREAMS")
    if streams:
        cfg["executor"]["streams"] = int(streams)
    return cfg
