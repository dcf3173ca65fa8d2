"""bench.py starts its own ranks: `--gpus N` without WORLD_SIZE spawns N processes and rank 0 prints one JSON line.
Rehearsed on the CPU (LDM_BENCH_FAKE=1: a stand-in sampler that is a function of the global chain index, gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LDM_BENCH_FAKE"] = "1"
    env.update(extra)
    return env


def _bench(*flags):
    env = _env()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *flags],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout      # ONE line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_starts_the_ranks_and_the_gathered_batch_is_world_size_invariant():
    one = _bench("--gpus", "1", "--global-batch", "6")
    two = _bench("--gpus", "2", "--global-batch", "6")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["scaling"] == two["scaling"] == "strong"
    assert two["config"]["global_batch"] == 6 and two["config"]["per_gpu_batch"] == 3
    assert one["gathered_digest"] == two["gathered_digest"]
    weak = _bench("--gpus", "2", "--batch", "3")           # 3 chains per rank: the same six global chains
    assert weak["scaling"] == "weak" and weak["config"]["global_batch"] == 6
    assert weak["gathered_digest"] == one["gathered_digest"]
    ragged = _bench("--gpus", "2", "--global-batch", "5")  # 3 + 2 chains
    assert ragged["config"]["per_gpu_batch"] == 3 and ragged["gathered_digest"] == _bench("--gpus", "1", "--global-batch", "5")["gathered_digest"]


def test_eight_self_launched_ranks_and_the_strong_scaling_shares_of_the_scaled_run():
    """The driver's N = 8 line and config 4's `--scaled --global-batch 128`: 16 chains per rank, and 100 chains as
    uneven shares (13 x 4 + 12 x 4); every gathered batch equals the single-process one and the line says how many
    ranks the process group saw."""
    one = _bench("--gpus", "1", "--scaled", "--global-batch", "128")
    eight = _bench("--gpus", "8", "--scaled", "--global-batch", "128")
    assert eight["n_gpus"] == 8 and eight["rccl_ranks"] == 8 and eight["collective_backend"] == "gloo"
    assert one["rccl_ranks"] is None
    assert eight["config"]["per_gpu_batch"] == 16 and eight["scaling"] == "strong"
    assert eight["gathered_digest"] == one["gathered_digest"]
    ragged = _bench("--gpus", "8", "--global-batch", "100")
    assert ragged["config"]["per_gpu_batch"] == 13
    assert ragged["gathered_digest"] == _bench("--gpus", "1", "--global-batch", "100")["gathered_digest"]
    weak = _bench("--gpus", "8", "--batch", "16")
    assert weak["scaling"] == "weak" and weak["config"]["global_batch"] == 128
    assert weak["gathered_digest"] == one["gathered_digest"]


def test_a_rank_that_dies_ends_the_run_promptly_with_its_status():
    """One rank exits after joining the group; the others wait in a collective.  The launcher must stop them and
    return the failing status instead of waiting for ever."""
    import time
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--gpus", "4",
                        "--batch", "2"], env=_env(LDM_BENCH_FAKE_FAIL_RANK="2"), capture_output=True, text=True, timeout=240)
    assert p.returncode == 3, (p.returncode, p.stderr[-1500:])
    assert "rank 2 exited with status 3" in p.stderr
    assert time.monotonic() - t0 < 120
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_the_launcher_deadline_stops_ranks_that_never_finish():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200000", "--warmup", "1", "--gpus", "2",
                        "--batch", "64"], env=_env(LDM_BENCH_DEADLINE_S="8"), capture_output=True, text=True, timeout=240)
    assert p.returncode == 124, (p.returncode, p.stderr[-1500:])
