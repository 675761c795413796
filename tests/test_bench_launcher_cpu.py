"""`python bench.py --gpus N` (N > 1) as a plain command starts its own ranks as a child process and hands their return
code on (CPU suite: without a GPU every rank stops with "needs an MI355X GPU" -- what is checked is that the ranks WERE
started through torch.distributed.run and that the failure is propagated, with no result line)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_gpus_2_spawns_ranks_and_propagates_their_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered on the GPU by tests/test_gpu_bench_contract.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dims", "16", "16", "512", "--steps", "1",
                        "--warmup", "0"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode(errors="replace")
    assert p.returncode != 0
    assert "needs an MI355X GPU" in err, err[-1500:]             # (the ranks ran bench.py's main)
    assert "launch with torch.distributed.run" not in err        # (and were not refused for a missing launcher)
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
