"""`python bench.py --gpus N` must start its own N ranks (the driver runs exactly that command line): the launcher path is
exercised here without a GPU through --dry-run (rendezvous over gloo, the deterministic tile partition, one all-gather)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_and_relays_one_line():
    p = _run("--gpus", "2", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    assert len(out["tiles_per_rank"]) == 2 and sum(out["tiles_per_rank"]) == out["tiles"] == 1600
    assert min(out["tiles_per_rank"]) > 600                      # both ranks got a real share


def test_failing_rank_gives_nonzero_exit_and_no_line():
    p = _run("--gpus", "2", "--dry-run", "--fail-rank", "1")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_refused():
    p = _run("--gpus", "8", "--dry-run", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "refusing" in p.stderr
    assert not p.stdout.strip()


def test_single_rank_dry_run():
    p = _run("--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip())
    assert out["n_gpus"] == 1 and out["tiles_per_rank"] == [1600]
