"""bench.py's launch contract on CPU (no GPU is touched: DEXSIM_BENCH_LAUNCH_PROBE makes every rank report and exit
before importing torch.cuda): `python bench.py --gpus N` without a launcher environment starts N ranks itself, one per
GPU index, over 127.0.0.1; a rank count that differs from --gpus is an error (round 1 silently ran a single rank)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(DEXSIM_BENCH_LAUNCH_PROBE="1", **env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_gpus_n_starts_n_ranks_itself():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stdout + r.stderr
    seen = sorted(re.findall(r"probe rank (\d) local_rank (\d) world (\d) master (\S+)", r.stdout + r.stderr))
    assert seen == [("0", "0", "2", "127.0.0.1"), ("1", "1", "2", "127.0.0.1")], r.stdout + r.stderr


def test_single_gpu_runs_in_process():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0 and "probe rank 0 local_rank 0 world 1" in r.stdout


def test_rank_count_mismatch_is_an_error():
    r = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
