"""Configuration.  The reference loads an AWS-centric YAML (numpywren/config.py:39-50); here
`default()` returns a local dict describing the single-node MI355X executor.  Keys can be
overridden with the environment variables listed below or a YAML file named by
$NUMPYWREN_AMD_CONFIG_FILE.
"""
import os

DEFAULTS = {
    "executor": {
        "streams": 4,            # HIP streams in the worker pool (job_runner pipeline_width analogue)
        "priority_stream": True,  # one high-priority stream for the critical path
        "exact_zero_shortcircuit": True,  # reproduce the reference's allclose(x, 0) early-outs
        "reclaim_intermediates": False,   # free intermediate tiles after their last reader
    },
    "store": {
        "tier": "hbm",           # "hbm" (device memory) or "host" (pinned/pageable host memory)
        "device_pool": True,
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
    streams = os.environ.get("NUMPYWREN_AMD_STREAMS")
    if streams:
        cfg["executor"]["streams"] = int(streams)
    return cfg
