"""Data-parallel sharding of independent requests over the GPUs of one node.

The reference shards inference by process fan-out over dataset chunks and merges JSONL files
(scripts/srgpt/eval/srgpt_bench.sh:9-45, llava/eval/eval_spatial.py:72-80: `split_list`/`get_chunk`, chunk
size ceil(len/n)).  Here: one rank per GPU under torchrun, the same contiguous ceil(len/n) chunking, and ONE
exchange step -- an all-gather of the generated ids over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in CPU tests).
No collective sits on the per-token path."""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("SRGPT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _parse_cpulist(text: str) -> List[int]:
    """sysfs cpulist ("0-63,128-191") -> sorted cpu ids"""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(out))


def gpu_numa_cpus(device_index: int, sysfs: str = "/sys") -> Optional[List[int]]:
    """cpus of the NUMA node the GPU hangs off (sysfs `local_cpulist` of its PCI function), or None when the platform does not say
    (no PCI ids from the runtime, numa_node == -1, containers without /sys/bus/pci)."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        base = os.path.join(sysfs, "bus", "pci", "devices", bdf)
        with open(os.path.join(base, "numa_node")) as f:
            if int(f.read().strip()) < 0:
                return None
        with open(os.path.join(base, "local_cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None


def plan_rank_affinity(local_rank: int, local_world: int, allowed: Sequence[int], numa_cpus: Optional[Sequence[int]] = None,
                       ranks_on_node: Optional[Sequence[int]] = None) -> List[int]:
    """The cpu set rank `local_rank` of `local_world` should run on: a DISJOINT share of the allowed cpus, taken from its GPU's NUMA
    node when that is known.  ranks_on_node: the local ranks whose GPUs share this NUMA node (they split its cpus between them);
    unknown topology: the allowed set is cut into local_world contiguous shares.  Never empty: falls back to the allowed set."""
    allowed = sorted(set(int(c) for c in allowed))
    if not allowed or local_world <= 1:
        return allowed
    pool, peers = allowed, list(range(local_world))
    if numa_cpus:
        near = [c for c in allowed if c in set(numa_cpus)]
        if near and ranks_on_node and local_rank in ranks_on_node and len(near) >= len(ranks_on_node):
            pool, peers = near, sorted(ranks_on_node)
    k, n = peers.index(local_rank) if local_rank in peers else local_rank % len(peers), len(peers)
    share = pool[k * len(pool) // n:(k + 1) * len(pool) // n]
    return share or allowed


def pin_rank_to_cores(local_rank: int, local_world: int, device_index: Optional[int] = None) -> dict:
    """Give this rank's process (and the threads it starts from here on) its own cpus next to its GPU.  N Python processes issue
    ~500 launches per request prefix each with < 0.1 ms of slack (profiles/r05_prefix_bs1.txt): left to the scheduler they migrate
    across sockets and share cores.  The reference's eval fan-out (scripts/srgpt/eval/srgpt_bench.sh:23-34) leaves placement to the
    OS; this is the MI355X-node version of it.  Returns what was done, for the bench line (config.dist)."""
    info = {"pinned": False, "local_rank": local_rank, "local_world": local_world}
    if not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return info
    try:
        allowed = sorted(os.sched_getaffinity(0))
        numa = gpu_numa_cpus(device_index) if device_index is not None and torch.cuda.is_available() else None
        peers = None
        if numa is not None:
            peers = [r for r in range(local_world) if gpu_numa_cpus(r % max(torch.cuda.device_count(), 1)) == numa]
        cpus = plan_rank_affinity(local_rank, local_world, allowed, numa, peers)
        os.sched_setaffinity(0, cpus)
        # one launcher thread + a helper or two: torch's intra-op pool must not spawn a thread per visible core
        torch.set_num_threads(max(1, min(len(cpus), 8)))
        info.update(pinned=True, cpus=f"{cpus[0]}-{cpus[-1]}" if cpus == list(range(cpus[0], cpus[-1] + 1)) else cpus[:64],
                    n_cpus=len(cpus), numa_known=numa is not None)
    except OSError as e:  # (a cgroup that forbids it: run unpinned and say so)
        info["error"] = str(e)
    return info


def barrier(local: Optional[int] = None, group=None) -> None:
    """dist.barrier that names this rank's device under the RCCL backend: without `device_ids` the nccl backend guesses the device
    from the global rank (and warns), which is wrong the moment ranks and devices are not numbered alike."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    if dist.get_backend(group) == "nccl":
        dev = torch.cuda.current_device() if local is None else local
        dist.barrier(group=group, device_ids=[dev])
    else:
        dist.barrier(group=group)


def split_list(lst: Sequence, n: int) -> List[Sequence]:
    """eval_spatial.py:72-75: chunks of size ceil(len/n) (the last ranks may get fewer / none)."""
    chunk = math.ceil(len(lst) / n) if len(lst) else 1
    return [lst[i:i + chunk] for i in range(0, len(lst), chunk)]


def get_chunk(lst: Sequence, n: int, k: int) -> Sequence:
    """eval_spatial.py:78-80, but ranks beyond the last chunk get an empty shard instead of IndexError."""
    chunks = split_list(lst, n)
    return chunks[k] if k < len(chunks) else lst[0:0]


def _gather_rows(local: torch.Tensor, pad, group=None) -> torch.Tensor:
    """All-gather 2-D per-rank blocks [rows_r, cols_r] -> [sum rows, max cols] in rank order.  Shapes may differ per rank
    (ragged shards, early EOS): sizes are exchanged first, payloads padded to the common shape -> one collective each."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    out_dev = local.device
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        local = local.cpu()  # gloo moves host buffers; RCCL ("nccl") takes the device tensor directly
    dev = local.device
    shape = torch.tensor(list(local.shape) if local.dim() == 2 else [0, 0], dtype=torch.int64, device=dev)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    mb = max(int(s[0]) for s in shapes)
    mg = max(int(s[1]) for s in shapes)
    buf = torch.full((mb, mg), pad, dtype=local.dtype, device=dev)
    if local.numel():
        buf[:local.shape[0], :local.shape[1]] = local
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    rows = [b[:int(s[0])] for b, s in zip(bufs, shapes)]
    return (torch.cat(rows, dim=0) if rows else buf[:0]).to(out_dev)


def gather_ids(local_ids: torch.Tensor, pad_id: int = 0, group=None) -> torch.Tensor:
    """All-gather generated ids [B_local, G_local] (int64) from every rank -> [sum B_local, max G] in rank order."""
    return _gather_rows(local_ids, pad_id, group)


def gather_logits(local_logits: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather last-position logits [B_local, V] (fp32) from every rank -> [sum B_local, V] in rank order: the exchange
    the north-star names.  V * 4 B per row (0.5 MB at V = 128258): one all-gather after the shard is done, never per token."""
    return _gather_rows(local_logits.float(), 0.0, group)


def generate_data_parallel(model, requests: Sequence[dict], group=None, pad_id: int = 0, **gen_kwargs) -> torch.Tensor:
    """Run `model.generate` on this rank's contiguous shard of `requests` (dicts of generate() kwargs) and
    all-gather the ids.  Every rank returns the ids of ALL requests in request order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = get_chunk(list(requests), world, rank)
    outs = [model.generate(**req, **gen_kwargs) for req in mine]
    dev = outs[0].device if outs else (model.device if hasattr(model, "device") else "cpu")
    if outs:
        g = max(o.shape[1] for o in outs)
        local = torch.full((sum(o.shape[0] for o in outs), g), pad_id, dtype=torch.int64, device=dev)
        r = 0
        for o in outs:
            local[r:r + o.shape[0], :o.shape[1]] = o
            r += o.shape[0]
    else:
        local = torch.zeros((0, 0), dtype=torch.int64, device=dev)
    return gather_ids(local, pad_id, group)


def forward_data_parallel(model, requests: Sequence[dict], group=None) -> torch.Tensor:
    """Run `model.forward` (llava_llama.py:100-192 surface) on this rank's shard and all-gather the LAST-position logits
    of every request: [len(requests), V] fp32 on every rank, request order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = get_chunk(list(requests), world, rank)
    rows = []
    for req in mine:
        out = model.forward(**req)
        logits = out.logits if hasattr(out, "logits") else out
        rows.append(logits[:, -1, :].float())
    if rows:
        local = torch.cat(rows, dim=0)
    else:
        dev = model.device if hasattr(model, "device") else "cpu"
        local = torch.zeros((0, 0), dtype=torch.float32, device=dev)
    return gather_logits(local, group)
