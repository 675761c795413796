"""bench.py prints ONE JSON line with the fields the driver reads (run on a small volume)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dims", "96", "128", "512", "--steps", "4", "--warmup", "1",
                        "--no-secondary"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel_launches_timed"] == 4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"] + 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert abs(d["value"] - 96 * 128 * 512 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]


def test_bench_two_ranks_rehearsal_runs_the_multi_gpu_path():
    """`bench.py --gpus 2` as the driver launches it, with two gloo ranks sharing this one GPU (TA_BENCH_BACKEND=gloo: RCCL
    cannot put two ranks on one device): cost-balanced slab cuts, reduce-scattered sums, two steps in flight, ONE line from rank 0."""
    env = dict(os.environ, TA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29650 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dims", "96", "64", "512",
                        "--steps", "3", "--warmup", "1", "--settle-ms", "5"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["per_label_reduce"] == "scatter"
    cuts = d["config"]["slab_cuts"]
    assert cuts[0] == 0 and cuts[-1] == 96 and len(cuts) == 3 and 0 < cuts[1] < 96
    assert d["config"]["steps_in_flight"] == 2 and d["roofline"]["frac"] > 0 and "global_adjacency_gather_ms" in d["secondary"]


def test_bench_gpus_2_as_a_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver starts the N = 1 bench): bench.py starts
    `torch.distributed.run` itself as a child process, relays rank 0's one JSON line and the return code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(TA_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dims", "96", "64", "512",
                        "--steps", "3", "--warmup", "1", "--settle-ms", "5"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["steps_in_flight"] == 2 and d["roofline"]["frac"] > 0
