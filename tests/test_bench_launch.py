"""CPU: `python bench.py --gpus N` starts itself the way the driver starts it -- bare, no launcher -- and must bring up one
rank per GPU under torch.distributed.run (127.0.0.1 rendezvous), keep the one-JSON-line contract on rank 0, and only then
fail for lack of a GPU.  `--dry-run` swaps RCCL + kernels for gloo so that the start-up path runs here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT,
                          env=e, timeout=timeout)


def test_bare_gpus2_self_launches_two_ranks():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1] and out["steps"] == 3 and out["allreduce_sum_ok"]


def test_under_a_launcher_with_matching_world_size():
    """the documented launch line: torch.distributed.run --nproc-per-node N bench.py --gpus N"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--dry-run"], capture_output=True, text=True, cwd=ROOT, env=e, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["ranks"] == [0, 1]


def test_world_size_mismatch_is_a_clear_error():
    r = _run("--gpus", "2", "--dry-run", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_without_a_gpu_the_real_run_stops_at_the_cuda_check():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "there is no CPU fallback" in r.stderr and "WORLD_SIZE" not in r.stderr.split("AssertionError")[-1]


def test_dry_run_world8_line_carries_the_n1_strings():
    """the N > 1 line must be mechanically comparable with the N = 1 line (SCALE vs BENCH): same `metric`, same `config.workload`,
    the contract's keys, global batch = 8 x the per-GPU batch"""
    sys.path.insert(0, ROOT)
    import bench
    r = _run("--gpus", "8", "--steps", "2", "--warmup", "0", "--dry-run", timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["metric"] == bench.METRIC and out["config"]["workload"] == bench.WORKLOAD
    assert out["n_gpus"] == 8 and out["ranks"] == list(range(8)) and out["config"]["global_batch"] == 8 * bench.PER_GPU_BATCH
    assert out["config"]["parallelism"] == "dp8" and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["unit"] == "frame-pairs/s" and out["allreduce_sum_ok"]
    # the strings are module constants used by the real (GPU) line too
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count('"metric": METRIC') >= 2 and src.count('"workload": WORKLOAD') >= 2
