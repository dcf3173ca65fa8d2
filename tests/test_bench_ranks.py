"""bench.py starts its own ranks: `--gpus N` without WORLD_SIZE spawns N processes and rank 0 prints one JSON line.
Rehearsed on the CPU (LDM_BENCH_FAKE=1: a stand-in sampler that is a function of the global chain index, gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LDM_BENCH_FAKE"] = "1"
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
