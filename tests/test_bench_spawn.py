"""CPU: `python bench.py --gpus N` without a launcher starts N fresh ranks itself and rank 0 reports n_gpus = N
(MUSTAFAR_BENCH_DRYRUN=1: the launch plumbing only -- ranks, gloo barrier, max-over-ranks, one JSON line; no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], MUSTAFAR_BENCH_DRYRUN="1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True


def test_launcher_environment_must_agree_with_the_flag():
    r = _run(["--gpus", "2"], MUSTAFAR_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_one_rank_needs_no_spawn():
    r = _run(["--gpus", "1"], MUSTAFAR_BENCH_DRYRUN="1")
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
