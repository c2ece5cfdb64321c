"""`bench.py` end to end on the CPU emulation of the kernels (`--emu`: tiny workload, gloo): the step function, the
self-launch of N workers from a plain invocation, gallery sharding + all-gather, the chunked / multi-layer configs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=600):
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emu", "--steps", "1", "--warmup", "0", *args],
                       capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_gpus2_invocation_starts_its_own_workers_and_matches_one_process():
    one = run_bench("--gallery-per-gpu", "12")
    two = run_bench("--gpus", "2", "--gallery-per-gpu", "6")  # no torchrun around it: bench.py launches its two ranks
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["parallelism"] == "gallery-shard x2"
    # the same 4 queries against the same 12 gallery items, sharded or not: identical ranks
    for key in ("rank1", "mAP", "mean_rank"):
        assert one[key] == two[key]
    assert two["value"] > 0 and two["unit"] == "pairs/s" and two["metric"].startswith("query x gallery NCC")
    # the N > 1 line proves what the collective saw: backend, ranks, timed all-gather, payload, and that every rank holds
    # the same gathered matrix (bench.py exits non-zero when the digests differ)
    assert "collective" not in one
    col = two["collective"]
    assert col["backend"] == "gloo" and col["world_seen"] == 2 and col["ranks_reporting"] == 2
    assert col["payload_bytes"] == 4 * 6 * 4 and col["gathered_bytes"] == 4 * 12 * 4
    assert col["allgather_ms"] > 0 and col["allgathers_timed"] == 1 and col["digest_agrees_on_all_ranks"] is True
    assert len(col["matrix_digest"]) == 16


@pytest.mark.parametrize("config", [3, 4, 5])
def test_other_configs_step(config):
    out = run_bench("--config", str(config), "--gallery-per-gpu", "5")
    assert out["config"]["workload"].startswith(f"config {config}:")
    assert out["config"]["storage"] == {3: "bfloat16", 4: "float16", 5: "float16"}[config]
    assert 0.0 < out["rank1"] <= 1.0
