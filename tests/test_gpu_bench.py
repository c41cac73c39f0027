"""bench.py end to end on the tiny plumbing model: the driver's contract (one JSON line, the required keys, `roofline` measured at
the workload's row count, `cpu_baseline` at N = 1) is exercised by the GPU suite, so a broken bench line cannot ship unnoticed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "tiny", "--steps", "2", "--warmup", "1",
                        "--max-new-tokens", "6", "--cpu-decode-steps", "2", *extra],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,rows,cpu", [((), 1, True), (("--batch", "4", "--no-cpu-baseline"), 4, False),
                                            (("--weights", "fp8", "--batch", "3", "--no-cpu-baseline"), 3, False),
                                            (("--no-graph", "--no-cpu-baseline"), 1, False)])
def test_bench_line_contract(extra, rows, cpu):
    d = _run(*extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - rows * 6 * 2 / (d["ms_per_step"] * 2e-3)) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["decode_rows"] == rows
    assert ("skinny" in r["kernel"]) == (rows > 2)
    assert "workload" in d["config"] and d["config"]["requests_per_step_per_gpu"] == rows
    if cpu:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "tokens/s"
    else:
        assert d["cpu_baseline"] is None


def test_bench_self_launch_two_ranks_on_one_device():
    """`python bench.py --gpus 2` forks its own ranks (one device here -> the ranks share it and rendezvous over gloo)."""
    d = _run("--gpus", "2", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["config"]["dist"]["world_size_seen"] == 2
    assert len(d["config"]["dist"]["per_rank_tokens_per_s"]) == 2
    assert d["config"]["parallelism"] == "dp2"
