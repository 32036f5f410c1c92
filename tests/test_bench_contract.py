"""bench.py's one-line contract (driver-facing): field names, types and the two extra objects.  The GPU test runs a
reduced workload (small tiles) through the real code path; the CPU test checks the committed round-1 line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
            "config": dict}


def _check_line(line, n_gpus=1):
    for k, t in REQUIRED.items():
        assert k in line and isinstance(line[k], t), k
    assert "vs_baseline" in line and line["vs_baseline"] is None      # BASELINE.md holds no published number
    assert line["n_gpus"] == n_gpus and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["dtype"] == "f64" and line["data"] == "synthetic" and "workload" in line["config"]
    assert line["value"] > 0 and line["ms_per_step"] > 0


def test_committed_round1_line():
    with open(os.path.join(ROOT, "profiles", "r01_bench_line.json")) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    _check_line(line)
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 78.6
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["traffic"] > 0
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert abs(line["value"] - line["steps"] * (line["config"]["n"] ** 3 / 3) / (line["ms_per_step"] * 1e-3 * line["steps"]) / 1e12) < 0.05


@pytest.mark.gpu
def test_bench_runs_and_prints_one_json_line():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--tile", "512",
           "--tiles", "4", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    _check_line(line)
    assert line["config"]["residual_all_tiles"] < 1e-13
    # every timed step's own duration (timing events between the steps' completion marks), its median, and the outliers
    assert len(line["step_ms"]) == 2 and all(t > 0 for t in line["step_ms"]) and line["ms_per_step_median"] > 0
    assert abs(sum(line["step_ms"]) / 2 - line["ms_per_step"]) < 0.5 * line["ms_per_step"] and isinstance(line["outliers"], list)
    assert line["roofline"]["traffic"] is None            # PMC figure only applies to the 4096^2 tile
    assert "north_star" not in line                       # only with the full-size tile


@pytest.mark.gpu
@pytest.mark.parametrize("workload,extra", [("tsqr", ["--leaves", "8"]), ("gemm32", ["--tiles", "2"])])
def test_bench_other_configs(workload, extra):
    """--workload tsqr / gemm32 (BASELINE.json configs[3] / [4]) through the same runner, reduced sizes."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--tile", "512",
           "--workload", workload] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    for k, t in REQUIRED.items():
        assert k in line and isinstance(line[k], t), k
    assert line["value"] > 0 and line["dtype"] == ("f64" if workload == "tsqr" else "f32")
    # both lines carry the two extra objects of the contract
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert roof["bound"] in ("mfma", "hbm") and (roof["unit"], roof["peak"]) in (("TFLOP/s", 78.6 if workload == "tsqr" else 157.3), ("GB/s", 8000.0))
    assert roof["achieved"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["launches"] > 0 and roof["avg_ms"] > 0
    assert "traffic" in roof and roof["algorithmic_flop_per_launch"] > 0
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == "TFLOP/s" and "extrapolated" in cpu["sample"]
    assert "MKL" in cpu["blas"]
    if workload == "tsqr":
        assert roof["mfma_bound_ms_per_batch_of_32"] > 0 and roof["bound"] == "mfma" and line["config"]["r_only_run"]["roofline"]["achieved"] > 0


def test_predicted_scaling_table_is_the_one_bench_quotes():
    """profiles/predicted_scaling.json (tools/predict_scaling.py) is the single source of the predicted 1 / 2 / 4 / 8-GPU
    figures: bench.py's N > 1 lines read it, and it covers the three multi-GPU workloads of BASELINE.json."""
    sys.path.insert(0, ROOT)
    import bench
    for workload in ("chol", "tsqr", "gemm32"):
        p = bench._predicted_scaling(workload)
        assert p is not None and "not a measurement" in p["source"]
        t = p["tflops_by_gpus"]
        assert set(t) == {"1", "2", "4", "8"} and t["1"] < t["2"] < t["4"] < t["8"]
        assert all(r["gpus"] in (1, 2, 4, 8) and r["ms"] > 0 for r in p["rows"])
    per_tile, src = bench._syrk_traffic()
    assert src.startswith("profiles/r") and 5.37e8 < per_tile < 1e10      # PMC passes of the newest round, above the algorithmic bytes


@pytest.mark.gpu
def test_bench_distributed_path_on_one_gpu():
    """The N > 1 code path of bench.py (dist.init_process_group -> RCCL transport -> lambdapack_run_distributed) with a
    world of one."""
    env = dict(os.environ, NUMPYWREN_AMD_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29661")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--tile", "512",
           "--tiles", "4", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["transport"] == "rccl" and line["value"] > 0
    # the evidence of who ran it: one rank joined, RCCL itself reports a communicator of one, one PCI device
    assert line["config"]["ranks_joined"] == 1 == line["config"]["rccl_nranks"] and len(line["config"]["devices"]) == 1
    assert line["config"]["devices"][0] and len(line["step_ms"]) == 1
    # first contact: the link calibration ran before the timed step (a world of one exchanges with itself over RCCL)
    cal, x = line["config"]["link_calibration"], line["config"]["xgmi_GBps"]
    assert cal["transport"] == "rccl" and "self-exchange" in cal["what"] and 0 < x["min"] <= x["median"] <= x["max"]
    assert cal["samples"] == [{"rank": 0, "d": 0, "GBps": x["min"]}]
    assert line["per_rank"][0]["timed_step"]["stall_reports"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("workload,extra,port", [("chol", ["--tiles", "4"], 29671), ("tsqr", ["--leaves", "8"], 29672),
                                                 ("gemm32", ["--tiles", "2"], 29673)])
def test_bench_two_ranks_on_one_gpu(workload, extra, port):
    """bench.py's N > 1 path end to end -- torchrun, ownership, owner-only input generation, the distributed executor,
    barrier + max-over-ranks timing, rank 0's single JSON line -- with two ranks sharing this box's one GPU (payloads
    staged through the host: RCCL refuses two ranks on a device; the RCCL transport itself is covered by
    tests/test_comm_gpu.py)."""
    env = dict(os.environ, NUMPYWREN_AMD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NUMPYWREN_AMD_STORE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--tile", "256",
           "--workload", workload] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    for k, t in REQUIRED.items():
        assert k in line and isinstance(line[k], t), k
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["transport"] == "host"
    # the new first-contact fields (host-staged numbers here: two ranks share the GPU) and the model re-evaluated at them
    x = line["config"]["xgmi_GBps"]
    assert 0 < x["min"] <= x["median"] <= x["max"] and line["config"]["link_calibration"]["transport"] == "host"
    pred = line["config"]["predicted_at_measured_link"]
    assert pred["gpus"] == 2 and pred["tflops"] > 0 and abs(pred["link_GBps_per_direction"] - x["median"]) < 0.01, pred
    # every N > 1 line explains itself: per rank the host's walk, its blocked time, how far the device ran behind, the bytes
    # moved (last timed step) and, from one extra step with executor.task_timers, device time by kernel and transport time
    assert len(line["per_rank"]) == 2
    for r, d in enumerate(line["per_rank"]):
        t, x = d["timed_step"], d["diagnostic_step"]
        assert t["rank"] == x["rank"] == r and t["transport"] == "host"
        for key in ("host_walk_ms", "host_blocked_ms", "drain_ms", "bytes_sent", "bytes_received", "positions", "tasks_run_here"):
            assert key in t and key in x, key
        assert x["kernel_busy_ms"] > 0 and x["kernel_ms_by_name"] and x["transfer_wait_ms"] >= 0 and x["step_ms"] > 0
    assert line["config"]["predicted"]["tflops_by_gpus"]["8"] > 0 and "not a measurement" in line["config"]["predicted"]["source"]
    # every N > 1 line carries its own measured N = 1 point: the same problem on rank 0 alone, before the communicator exists
    a = line["config"]["one_gpu_anchor"]
    assert a["tflops"] > 0 and a["ms_per_step"] > 0 and len(a["step_ms"]) == 2 and "one GPU" in a["what"]
    if workload == "chol":
        assert line["scaling"] == "strong" and sum(d["timed_step"]["bytes_sent"] for d in line["per_rank"]) > 0


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher: the script becomes the launcher, two ranks join the
    control group, and rank 0 reports exactly those two.  --dry-run stops after the rendezvous, so this runs without a GPU
    (gloo control group, host transport)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"]
    out = subprocess.run(cmd, cwd=ROOT, env=_clean_env(NUMPYWREN_AMD_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["dry_run"] is True and line["n_gpus"] == 2 == line["ranks_joined"] == len(line["ranks"])
    assert sorted(r["rank"] for r in line["ranks"]) == [0, 1] and sorted(r["local_rank"] for r in line["ranks"]) == [0, 1]
    assert len({r["pid"] for r in line["ranks"]}) == 2          # one process per rank


@pytest.mark.parametrize("world", ["1", "4"])
def test_bench_refuses_a_job_of_another_size(world):
    """--gpus N inside a job whose WORLD_SIZE is not N never prints a line (it used to run the one-GPU workload and label
    it n_gpus = N)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"]
    out = subprocess.run(cmd, cwd=ROOT, env=_clean_env(WORLD_SIZE=world, RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=" + world in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_step_stats_flags_a_stalled_step(capsys):
    sys.path.insert(0, ROOT)
    import bench
    s = bench.step_stats([26.5, 26.7, 80.0, 26.6, 26.4], 37.24)
    assert s["ms_per_step_median"] == 26.6 and s["outliers"] == [2] and s["step_ms"][2] == 80.0
    assert "differ" in capsys.readouterr().err
    assert bench.step_stats([26.5, 26.6], 26.55)["outliers"] == [] and capsys.readouterr().err == ""


@pytest.mark.gpu
def test_bench_two_ranks_without_a_launcher_on_one_gpu():
    """The same self-launch end to end on the GPU box: `python bench.py --gpus 2` with no WORLD_SIZE runs the two-rank
    Cholesky (ranks share the box's one GPU: host-staged payloads, allowed knowingly through NUMPYWREN_AMD_DIST_BACKEND)
    and the line carries the evidence of who ran it plus rank 0's one-GPU anchor of the same matrix."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tile", "256", "--tiles", "4"]
    out = subprocess.run(cmd, cwd=ROOT, env=_clean_env(NUMPYWREN_AMD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 == line["config"]["ranks_joined"] == len(line["per_rank"]) == len(line["config"]["devices"])
    assert line["config"]["transport"] == "host" and line["config"]["rccl_nranks"] is None
    assert len(line["step_ms"]) == 2 and line["ms_per_step_median"] > 0 and line["outliers"] in ([], [0], [1])
    a = line["config"]["one_gpu_anchor"]
    assert a["tflops"] > 0 and a["ms_per_step"] > 0 and len(a["step_ms"]) == 2
