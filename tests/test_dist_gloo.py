"""CPU, world_size 2 over gloo: request sharding + the all-gather exchange step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel:
    """generate() returns ids derived from the request so that ordering can be checked."""
    device = "cpu"

    def generate(self, input_ids=None, n_new=3, **kw):
        base = int(input_ids[0, 0])
        return torch.arange(base, base + n_new, dtype=torch.int64)[None]

    def forward(self, input_ids=None, **kw):
        base = float(input_ids[0, 0])
        return (torch.arange(3 * 7, dtype=torch.float32).reshape(1, 3, 7) + base)  # [B=1, T=3, V=7]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from spatialrgpt_amd.dist import forward_data_parallel, gather_ids, generate_data_parallel, init_distributed

    r, w, _ = init_distributed("gloo")
    assert (r, w) == (rank, world)
    # ragged shards: rank 0 has 2 rows x 4 ids, rank 1 has 1 row x 2 ids
    local = torch.full((2, 4), 10, dtype=torch.int64) if rank == 0 else torch.full((1, 2), 20, dtype=torch.int64)
    allids = gather_ids(local, pad_id=-1)
    assert allids.shape == (3, 4)
    assert allids[:2].eq(10).all() and allids[2, :2].eq(20).all() and allids[2, 2:].eq(-1).all()
    reqs = [dict(input_ids=torch.tensor([[100 * i]]), n_new=2 + (i % 2)) for i in range(5)]
    out = generate_data_parallel(_FakeModel(), reqs, pad_id=-1)
    exp = torch.full((5, 3), -1, dtype=torch.int64)
    for i in range(5):
        n = 2 + (i % 2)
        exp[i, :n] = torch.arange(100 * i, 100 * i + n)
    assert torch.equal(out, exp), (out, exp)
    # last-position logits of every request on every rank, request order (5 requests over 2 ranks: 3 + 2)
    lg = forward_data_parallel(_FakeModel(), [dict(input_ids=r["input_ids"]) for r in reqs])
    assert lg.shape == (5, 7) and lg.dtype == torch.float32
    for i in range(5):
        assert torch.equal(lg[i], torch.arange(14, 21, dtype=torch.float32) + 100 * i)
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


@pytest.mark.timeout(120)
def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_bench_self_launch_plumbing():
    """`python bench.py --gpus 2` must fork its own ranks when no launcher set WORLD_SIZE (the driver's command shape;
    reference pattern: scripts/srgpt/eval/srgpt_bench.sh:23-34) -- checked without a model through --selftest-launcher:
    rendezvous on 127.0.0.1, the ragged-safe id gather, exactly one rank-0 JSON line, exit code 0."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launcher", "--batch", "3"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world_size_seen"] == 2 and j["gathered_rows"] == 6 and j["launcher"] == "self"


@pytest.mark.timeout(300)
@pytest.mark.parametrize("preset,batch,weights", [("config2", 4, "bf16"), ("config4", 8, "fp8_w8a8")])
def test_bench_eight_rank_dry_run_names_the_global_batch(preset, batch, weights):
    """Nobody can hand this build 8 GPUs (one per lease; SCALE_rNN.json has been a `skipped` record every round), so the 8-rank leg
    of `bench.py --gpus 8 --preset config2|config4` is exercised with 8 CPU ranks over gloo: self-launch of 8 processes, rendezvous,
    the id gather over 8 shards, the per-rank rows, and a rank-0 line that names BASELINE configs[2] / configs[4]'s GLOBAL batch
    (32 / 64) and the mode the preset selects (configs[4] = fp8 weights on the fp8 matrix pipe = fp8_w8a8)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--preset", preset, "--selftest-launcher",
                        "--max-new-tokens", "16"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["world_size_seen"] == 8 and j["parallelism"] == "dp8" and j["launcher"] == "self"
    assert j["requests_per_step_per_gpu"] == batch and j["global_batch"] == 8 * batch == {"config2": 32, "config4": 64}[preset]
    assert j["gathered_rows"] == 8 * batch and j["llm_weights"] == weights
    assert f"configs[{preset[-1]}]" in j["workload"]
    assert j["per_rank_rows"] == [[r_, batch * 16] for r_ in range(8)]


def test_rank_affinity_plan_gives_disjoint_shares_next_to_the_gpu():
    """spatialrgpt_amd.dist.plan_rank_affinity (bench.py pins every rank with it): disjoint shares that cover the allowed cpus; with a
    known topology the ranks of one NUMA node split THAT node's cpus; never an empty set."""
    from spatialrgpt_amd.dist import _parse_cpulist, plan_rank_affinity

    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(256))
    shares = [plan_rank_affinity(r, 8, allowed) for r in range(8)]
    assert sorted(c for s in shares for c in s) == allowed and all(len(s) == 32 for s in shares)
    # two sockets: GPUs 0-3 on node 0 (cpus 0-63 + their hyperthreads 128-191), 4-7 on node 1
    node0 = list(range(0, 64)) + list(range(128, 192))
    node1 = [c for c in allowed if c not in node0]
    s = [plan_rank_affinity(r, 8, allowed, node0 if r < 4 else node1, [0, 1, 2, 3] if r < 4 else [4, 5, 6, 7]) for r in range(8)]
    assert all(set(s[r]) <= set(node0 if r < 4 else node1) for r in range(8))
    assert sorted(c for x in s for c in x) == allowed
    # a cgroup that exposes 8 cpus to 8 ranks: one each; fewer cpus than ranks: shares may repeat but are never empty
    assert [plan_rank_affinity(r, 8, list(range(8))) for r in range(8)] == [[c] for c in range(8)]
    assert all(plan_rank_affinity(r, 8, [3, 4]) for r in range(8))
    assert plan_rank_affinity(0, 1, allowed) == allowed
