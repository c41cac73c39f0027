#!/usr/bin/env python
"""bench.py -- region-grounded output tokens/sec of the MI355X path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1 works both ways: under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (the
  ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env) and as a plain `python bench.py --gpus N`, which forks its
  own N ranks -- one process per GPU, like the reference's launcher (scripts/srgpt/eval/srgpt_bench.sh:23-34) -- over RCCL.

One "step" = one whole request pass per rank: synthetic 384x384 image + depth map (SigLIP-so400m geometry; the
"336 px" of the metric name is not reachable with the shipped tower), 8 region masks, a
64-id prompt with one <image> and 8 (<mask>,<depth>) pairs, greedy decode of 128 new tokens (EOS disabled):
both ViT passes, feature refinement, mask pooling, projector, splice, prefill (T = 259) and 127 decode steps.
Weights: seeded random weights of the VILA1.5-8B (Llama-3-8B + SigLIP-so400m) architecture, bf16, generated on
the device; inputs are resident in HBM before the timed region.  Weak scaling: every rank serves its own
request stream (data parallel); the only exchange is an all-gather of the new ids.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the decode weight-streaming GEMV) and, at
N = 1, `cpu_baseline` (the oracle = CPU restatement of the reference, timed on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="vila15_8b", choices=["vila15_8b", "vila15_8b_clip336", "llama2_7b", "sheared_3b", "tiny"])
    ap.add_argument("--regions", type=int, default=8)
    ap.add_argument("--prompt-len", type=int, default=64)
    ap.add_argument("--max-new-tokens", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pin-as", default=None, metavar="RANK/WORLD",
                    help="measurement aid: pin this single process to the cpus local rank RANK of WORLD would get (scripts/round6/sibling_load.py)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--batch", type=int, default=1, help="equal-length requests per generate() call per GPU")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8", "fp8_w8a8"],
                    help="fp8 = weight-only OCP e4m3fn for the streamed LLM matrices, W8A16 throughout; fp8_w8a8 = BASELINE configs[4] "
                         "('fp8 weights on CDNA4 fp8 MFMA'): the same weights, prefill on the fp8 matrix pipe with per-token e4m3 "
                         "activations; neither is the headline (configs[1], bf16)")
    ap.add_argument("--preset", default=None, choices=["config1", "config2", "config3", "config4", "config4_w8a16"],
                    help="BASELINE.json configs[i] per-GPU shape: config1 = bs 1 (default); config2 = bs 32 over 8 GPUs = 4 requests "
                         "per GPU; config3 = llama2_7b, 16 regions, 512-id prompt; config4 = 'fp8 weights on CDNA4 fp8 MFMA', bs 64 over "
                         "8 GPUs = 8 per GPU = --weights fp8_w8a8 (the mode README / DESIGN quote for configs[4]); config4_w8a16 = the "
                         "same shape with W8A16 throughout (--weights fp8)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="no model, no GPU: the N ranks only rendezvous (gloo), exchange fake ids through the same gather and print "
                         "the rank-0 line -- checks the self-launch / rendezvous / gather plumbing on a CPU box")
    ap.add_argument("--cpu-decode-steps", type=int, default=12, help="decode steps the CPU baseline measures (full depth)")
    ap.add_argument("--no-cpu-fp32", action="store_true", help="skip the float32 leg of the CPU baseline (BASELINE.md asks for fp32 and bf16)")
    a = ap.parse_args()
    if a.preset == "config2":
        a.batch = 4
    elif a.preset == "config3":
        a.model, a.regions, a.prompt_len = "llama2_7b", 16, 512
    elif a.preset == "config4":
        a.weights, a.batch = "fp8_w8a8", 8
    elif a.preset == "config4_w8a16":
        a.weights, a.batch = "fp8", 8
    return a


def workload_name(args, cfg, T):
    geo = {"vila15_8b": "SpatialRGPT-VILA1.5-8B geometry (Llama-3-8B 32L/4096/GQA-8 + SigLIP-so400m 384px x2 passes + regiongpt "
                        "extractor + mlp_downsample)",
           "vila15_8b_clip336": "VILA1.5-8B LLM geometry behind the CLIP-L/14-336 tower (the only true 336-px tower: 577 tokens, "
                                "select_feature=patch -> 576)",
           "llama2_7b": "SpatialRGPT llama2_7b geometry (Llama-2-7B 32L/4096/MHA-32 + SigLIP-so400m 384px x2 passes)",
           "sheared_3b": "SpatialRGPT sheared_3b geometry (Sheared-LLaMA-2.7B 32L/2560/MHA-20 + SigLIP-so400m)", "tiny": "tiny plumbing model"}
    if args.model == "vila15_8b" and args.weights == "bf16" and args.batch == 1 and args.regions == 8:
        tag = "BASELINE configs[1]"
    elif args.model == "vila15_8b" and args.weights == "bf16" and args.batch == 4 and args.regions == 8:
        tag = "BASELINE configs[2] per-GPU shape (bs 32 over 8 GPUs = 4 requests per GPU)"
    elif args.model == "llama2_7b" and args.regions == 16 and args.prompt_len == 512:
        tag = "BASELINE configs[3]"
    elif args.model == "vila15_8b" and args.weights in ("fp8", "fp8_w8a8") and args.batch == 8:
        tag = "BASELINE configs[4] per-GPU shape (fp8 LLM weights, bs 64 over 8 GPUs = 8 requests per GPU)"
        if args.weights == "fp8_w8a8":
            tag += ", W8A8 prefill on the fp8 matrix pipe"
    else:
        tag = "non-BASELINE variant"
    return (f"{tag}: {geo[args.model]}, {args.regions} region masks, bs={args.batch} per GPU, prompt {args.prompt_len} ids -> "
            f"T={T}, greedy {args.max_new_tokens} new tokens")


def spliced_len(cfg, prompt_len):
    """T of the spliced stream: the prompt's ids minus the <image> sentinel plus the projector's tokens -- mlp_downsample halves
    the tower grid (27 -> 14: 196 tokens for SigLIP-384; 24 -> 12: 144 for CLIP-L/14-336)."""
    g = (cfg.grid + 1) // 2
    return prompt_len - 1 + g * g


def kernel_source_sha256():
    """sha256 over the sources of the decode GEMV (gemv.hip, common.h): a PMC summary under profiles/ belongs to ONE kernel source;
    bench.py only quotes its traffic figure when the hash recorded in it is the hash of the tree that is running."""
    import hashlib

    h = hashlib.sha256()
    for f in ("gemv.hip", "common.h"):
        with open(os.path.join(ROOT, "spatialrgpt_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def make_cfg(name):
    from spatialrgpt_amd.config import SrgptConfig

    if name == "tiny":
        return SrgptConfig(vit_hidden=64, vit_inter=176, vit_layers=3, vit_heads=4, image_size=378, hidden=64, inter=160,
                           layers=2, heads=4, kv_heads=2, vocab=128, mask_token_id=120, depth_token_id=121)
    if name == "vila15_8b_clip336":
        return SrgptConfig.clip_l14_336()
    return getattr(SrgptConfig, name)()


def synth_request(cfg, regions, prompt_len, seed, device, dtype):
    """SURVEY 8d synthetic request, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    S = cfg.image_size
    images = torch.randn((1, 3, S, S), generator=g, device=device).clamp_(-1, 1).to(dtype)
    depths = torch.randn((1, 1, S, S), generator=g, device=device).clamp_(-1, 1).expand(1, 3, S, S).contiguous().to(dtype)
    gc = torch.Generator().manual_seed(seed)
    m = torch.zeros((regions, S, S))
    for r in range(regions):
        hh, ww = [int(torch.randint(S // 8, S // 2 + 1, (1,), generator=gc)) for _ in range(2)]
        y0, x0 = int(torch.randint(0, S - hh + 1, (1,), generator=gc)), int(torch.randint(0, S - ww + 1, (1,), generator=gc))
        m[r, y0:y0 + hh, x0:x0 + ww] = 1.0
    masks = [m.to(device=device, dtype=dtype)]
    n_text = prompt_len - 2 - 2 * regions
    hi = min(cfg.mask_token_id, cfg.depth_token_id, cfg.vocab)
    txt = torch.randint(3, hi, (n_text,), generator=gc).tolist()
    pa = max(1, n_text // 4)
    per = max(1, (n_text - pa) // (regions + 1))
    seq = [1] + txt[:pa] + [-200]
    cur = pa
    for r in range(regions):
        seq += txt[cur:cur + per] + [cfg.mask_token_id, cfg.depth_token_id]
        cur += per
    seq += txt[cur:]
    assert len(seq) == prompt_len
    return torch.tensor([seq], device=device), images, depths, masks


CPU_THREADS = 16  # probed on the bench host class (scripts/cpu_probe.py): torch's CPU bf16 GEMMs are fastest at 16 threads


def cpu_baseline(cfg, sd_cpu, regions, prompt_len, g_cpu, dtype=torch.bfloat16):
    """The oracle (CPU restatement of the reference's path, oracle/srgpt_oracle.py) on the host cores at FULL depth and true
    widths: both tower passes, refinement / pooling / projector / splice, the T-position prefill with lm_head on every row (as the
    reference computes it) and `g_cpu` greedy decode steps -- every stage of the request is measured, only the decode-step count
    is bounded (the per-step cost is flat over 128 steps: the context grows from T to T+128).  Checker/baseline only."""
    from oracle import srgpt_oracle as so

    names = so.SrgptConfig.__dataclass_fields__
    ocfg = so.SrgptConfig(**{k: v for k, v in cfg.to_dict().items() if k in names})
    # thread count: probed on the 256-core bench host (scripts/cpu_probe.py) -- torch's CPU bf16 path is fastest
    # at 16 threads (22.9 ms / 2-layer decode step) and collapses beyond 64 (5.1 s at 256 threads)
    torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))
    ids, images, depths, masks = so.synth_inputs(ocfg, batch=1, regions=regions, prompt_len=prompt_len, seed=1, dtype=dtype)
    t = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        tower = so.vit_forward(sd_cpu, ocfg, torch.cat([images, depths], 0))
        t["vit_x2"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        hres, lres = so.feature_refinement(sd_cpu, tower[:1])
        me, de = so.region_extractor(sd_cpu, hres, tower[1:], masks)
        feats = so.mm_projector(sd_cpu, lres)
        emb, am, pid = so.splice(sd_cpu, ocfg, ids, None, feats, me, de, True)
        t["region_proj_splice"] = time.perf_counter() - t0
        kv = so.KVCache(ocfg.layers)
        T = emb.shape[1]
        t0 = time.perf_counter()
        logits = so.llama_forward(sd_cpu, ocfg, emb, torch.arange(T)[None], kv)
        t["prefill"] = time.perf_counter() - t0
        nxt = logits[:, -1].argmax(-1)
        t0 = time.perf_counter()
        for s in range(g_cpu):
            e = torch.nn.functional.embedding(nxt[:, None], sd_cpu["llm.model.embed_tokens.weight"])
            logits = so.llama_forward(sd_cpu, ocfg, e, torch.tensor([[T + s]]), kv, last_only=True)
            nxt = logits[:, -1].argmax(-1)
        t["decode_step"] = (time.perf_counter() - t0) / max(1, g_cpu)
    return t


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: fork N ranks of this script (one process per GPU, the reference's
    srgpt_bench.sh pattern), rendezvous on 127.0.0.1, pass rank 0's JSON line through.  Exit code = worst rank's."""
    import socket
    import subprocess

    with socket.socket() as so_:
        so_.bind(("127.0.0.1", 0))
        port = so_.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SRGPT_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll all ranks: the first one that dies takes the others with it (a survivor would sit in the rendezvous / all_reduce
    # until the backend's timeout), exit code = the worst rank's
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r_ = p.poll()
            if r_ is None:
                continue
            live.remove(p)
            rc = max(rc, abs(r_))
            if r_ != 0:
                for q in live:
                    q.kill()
    return rc


def launcher_selftest(args):
    """plumbing check of the N>1 leg without a model: rendezvous, barrier, ragged-safe id gather, one rank-0 JSON line."""
    import torch.distributed as dist

    from spatialrgpt_amd.dist import gather_ids, init_distributed

    from spatialrgpt_amd.dist import barrier

    rank, world, _ = init_distributed(backend="gloo")
    assert world == args.gpus
    G = args.max_new_tokens
    ids = torch.full((args.batch, G), rank, dtype=torch.int64)
    out = gather_ids(ids)
    barrier()
    assert out.shape == (world * args.batch, G) and all(int(out[r * args.batch, 0]) == r for r in range(world))
    # per-rank figures travel like the real run's (all_gather of one row per rank)
    mine = torch.tensor([float(rank), float(args.batch * G)], dtype=torch.float64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    if rank == 0:
        cfg = make_cfg(args.model)
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "world_size_seen": world, "gathered_rows": int(out.shape[0]),
                          "launcher": "self" if os.environ.get("SRGPT_BENCH_SELF_LAUNCHED") else "external",
                          "preset": args.preset, "llm_weights": args.weights, "requests_per_step_per_gpu": args.batch,
                          "global_batch": world * args.batch, "parallelism": f"dp{world}",
                          "workload": workload_name(args, cfg, spliced_len(cfg, args.prompt_len)),
                          "per_rank_rows": [[int(t[0]), int(t[1])] for t in allr]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.selftest_launcher:
        return launcher_selftest(args)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    from spatialrgpt_amd.dist import gather_ids, init_distributed

    ndev = torch.cuda.device_count()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    local_env = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    # fewer devices than ranks (a 1-GPU box asked for --gpus 2): ranks share devices round-robin; RCCL refuses two ranks on one
    # device, so that case rendezvous over gloo (the exchange is 1 KB of ids per rank) -- said so in the JSON line
    shared = world_env > ndev
    device = torch.device("cuda", local_env % ndev)
    torch.cuda.set_device(device)
    rank, world, local = init_distributed(backend="gloo" if shared else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    import torch.distributed as dist

    # every rank gets its own cpus next to its GPU (one node: LOCAL_WORLD_SIZE ranks share the host); reported in config.dist
    from spatialrgpt_amd.dist import pin_rank_to_cores
    if args.pin_as:
        pr, pw = (int(v) for v in args.pin_as.split("/"))
        affinity = pin_rank_to_cores(pr, pw, device.index)
    else:
        affinity = pin_rank_to_cores(local_env, int(os.environ.get("LOCAL_WORLD_SIZE", str(world_env))), device.index)

    from spatialrgpt_amd import _lib as L
    from spatialrgpt_amd import ops
    from spatialrgpt_amd.model import LlavaLlamaModel
    from spatialrgpt_amd.weights import synth_state_dict

    cfg = make_cfg(args.model)
    dtype = torch.bfloat16
    G = args.max_new_tokens
    t_build = time.perf_counter()
    sd = synth_state_dict(cfg, seed=0, dtype=dtype, device=device)
    # CPU baseline runs the SAME weights at full depth: copy them to the host before the engine consumes them
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
    # RoPE table: the request's positions (spliced prompt + new tokens), at least the 1024 every earlier round's line used
    rope_positions = max(1024, (spliced_len(cfg, args.prompt_len) + G + 127) // 128 * 128)
    model = LlavaLlamaModel(cfg, sd, device=device, dtype=dtype, rope_positions=rope_positions, consume_state_dict=True,
                            llm_weight_format=args.weights if args.weights != "bf16" else "native")
    del sd
    model.engine.use_graph = not args.no_graph
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    reqs = [synth_request(cfg, args.regions, args.prompt_len, 1 + rank * args.batch + i, device, dtype) for i in range(args.batch)]
    req = (torch.cat([r[0] for r in reqs], 0), torch.cat([r[1] for r in reqs], 0), torch.cat([r[2] for r in reqs], 0),
           [r[3][0] for r in reqs])

    def step():
        ids = model.generate(req[0], images=req[1], depths=req[2], masks=req[3], do_sample=False, max_new_tokens=G,
                             eos_token_id=None)
        return gather_ids(ids) if world > 1 else ids  # the one exchange step of the DP path

    for _ in range(args.warmup):
        out = step()

    from spatialrgpt_amd.dist import barrier as dist_barrier

    def barrier():
        torch.cuda.synchronize()
        dist_barrier(device.index)  # names this rank's device under RCCL (no guessing from the global rank)
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.shape[-1] == G and out.shape[0] == world * args.batch, out.shape
    tokens = world * args.steps * G * args.batch
    value = tokens / dt
    dist_info = {"launcher": "self (bench.py forked its ranks)" if os.environ.get("SRGPT_BENCH_SELF_LAUNCHED") else
                 ("torch.distributed.run" if world > 1 else "single process"), "world_size_seen": world, "affinity_rank0": affinity}
    if world > 1:
        mine = torch.tensor([args.steps * G * args.batch / dt_local, float(device.index)], dtype=torch.float64,
                            device=device if dist.get_backend() == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dist_info.update(backend=("rccl (torch backend 'nccl')" if dist.get_backend() == "nccl" else dist.get_backend()),
                         per_rank_tokens_per_s=[round(float(t[0]), 2) for t in allr], rank_devices=[int(t[1]) for t in allr],
                         devices_visible=ndev, ranks_share_devices=bool(shared))

    # ---------------- roofline of the dominant kernel: decode weight-streaming GEMV (gate/up + SwiGLU) --------
    # measured live with events on the launch stream, cycling over all layers' matrices (7.5 GB >> 256 MB L3)
    eng = model.engine
    roof = None
    if rank == 0:
        # the product as the timed workload launches it: one activation row per request of the batch (one row: the VALU GEMV,
        # 2+ rows: the MFMA skinny kernel)
        rows = args.batch
        x = torch.randn((rows, cfg.hidden), device=device).to(dtype)
        outb = torch.empty((args.batch, cfg.inter), device=device, dtype=dtype)
        fp8 = args.weights != "bf16"
        wgu = eng.w.llm_q["wgu"][0] if fp8 else eng.w.llm_t["wgu"]
        wsc = eng.w.llm_q["wgu"][1] if fp8 else [None] * len(wgu)

        # 2+ rows: as the batched decode step launches it -- RMSNorm statistics from the published table (srgpt_gemv_rowss), the
        # packed (MFMA-operand-order) copy of the fp8 matrix when the engine holds one
        use_rowss = rows > 1 and ops.gemv_rowss_supported(rows, fp8)
        pk = getattr(eng.w, "llm_pk", {}).get("wgu") if fp8 else None
        table = torch.zeros((rows, L.ROWSS_STRIDE), device=device)
        table[:, 0] = x.float().pow(2).sum(-1)

        def gateup(i):
            nw = eng.w.llm_t["mlp_norm"][i]
            if use_rowss:
                if pk is not None:
                    kw = dict(w8=pk[i], wscale=wsc[i], packed_rows=eng.w.pk_rows["wgu"], n_rows=wgu[i].shape[0])
                else:
                    kw = dict(w8=wgu[i], wscale=wsc[i]) if fp8 else dict(w=wgu[i])
                ops.gemv_rowss(x, norm_w=nw, eps=cfg.rms_eps, swiglu=True, out=outb, rowss_in=table, **kw)
            elif fp8:
                ops.gemv_w8(x, wgu[i], wsc[i], norm_w=nw, eps=cfg.rms_eps, swiglu=True, out=outb)
            else:
                ops.gemv(x, wgu[i], norm_w=nw, eps=cfg.rms_eps, swiglu=True, out=outb)

        for i in range(len(wgu)):  # warm (JIT-free, but first-touch TLB)
            gateup(i)
        reps = 3
        evs = []
        for _ in range(reps):
            for i in range(len(wgu)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gateup(i)
                e1.record()
                evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        avg_ms = sum(ms) / len(ms)
        alg_bytes = wgu[0].numel() * wgu[0].element_size()  # every weight byte exactly once per launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # whole decode phase (graph replays incl. attention + launch gaps), same events technique
        st, _, _ = eng.prefill(torch.randn((args.batch, spliced_len(cfg, args.prompt_len), cfg.hidden), device=device).to(dtype), max_new=G)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        eng.greedy_decode(st, G)
        e1.record()
        torch.cuda.synchronize()
        dec_ms = e0.elapsed_time(e1) / G
        # the request prefix (both ViT passes, refinement, pooling, projector, splice, prefill, first id): whole requests of ONE new token
        pre = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            model.generate(req[0], images=req[1], depths=req[2], masks=req[3], do_sample=False, max_new_tokens=1, eos_token_id=None)
            e1.record()
            torch.cuda.synchronize()
            pre.append(e0.elapsed_time(e1))
        prefix_ms = min(pre[1:])  # the first call re-captures nothing but warms the 1-token buffers
        wbytes = eng.w.llm_weight_bytes()
        # HBM bytes per launch: PMC counters cannot be read from inside this process -- the figure comes from the tracked
        # summary of a separate `rocprofv3 --pmc FETCH_SIZE` pass over this same command (x2 gfx950 correction), and says so
        traffic, traffic_src = None, None
        pmcs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_gemv.json"))
        if args.model == "vila15_8b" and pmcs and not fp8 and rows == 1:
            pmc = os.path.join(ROOT, "profiles", pmcs[-1])
            rec = json.load(open(pmc))
            if rec.get("kernel_source_sha256") == kernel_source_sha256():
                traffic = rec.get("traffic_bytes_per_launch")
                traffic_src = (f"static: {os.path.relpath(pmc, ROOT)} (separate rocprofv3 --pmc FETCH_SIZE pass over this kernel source, "
                               f"sha256 {rec['kernel_source_sha256'][:12]}; x2 gfx950 correction; not measured in this run)")
            else:
                traffic_src = (f"none: {os.path.relpath(pmc, ROOT)} was recorded for another kernel source (sha256 "
                               f"{str(rec.get('kernel_source_sha256'))[:12]} != {kernel_source_sha256()[:12]}); re-run scripts/profile_round.sh")
        kname = (("skinny_kernel<swiglu, W8" + (", packed>" if pk is not None else ">") if rows > 1 else "gemv_w8_kernel<swiglu>") if fp8 else
                 ("skinny_kernel<swiglu>" if rows > 1 else f"gemv_kernel<bf16,{rows},swiglu>"))
        roof = {"bound": "hbm", "kernel": f"{kname} (decode gate/up projection at {rows} activation row(s), 54% of streamed bytes"
                                          + (", fp8 weights)" if fp8 else ")"),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src, "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 5),
                "how": "hip events around each launch on the launch stream, 3 sweeps over the 32 layers' matrices (cold in L3)",
                "decode_ms_per_step": round(dec_ms, 4), "decode_ms_per_token": round(dec_ms / rows, 4), "decode_rows": rows, "decode_weight_bytes_per_token": wbytes,
                "decode_hbm_gbs_whole_step": round(wbytes / (dec_ms * 1e-3) / 1e9, 1),
                "decode_frac_whole_step": round(wbytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "prefix_ms_per_call": round(prefix_ms, 3),
                "prefix_how": "hip events around generate(max_new_tokens=1) with this run's requests: vision x2, refinement, pooling, projector, splice, prefill, first id (best of 3)"}

    cpu = None
    if sd_cpu is not None:
        g_cpu = max(1, args.cpu_decode_steps)
        t = cpu_baseline(cfg, sd_cpu, args.regions, args.prompt_len, g_cpu)
        total = t["vit_x2"] + t["region_proj_splice"] + t["prefill"] + (G - 1) * t["decode_step"]
        # fp32 leg (BASELINE.md section 3 asks for fp32 and bf16): the same request on the same (bf16-valued) weights in float32,
        # bounded harder -- 3 decode steps -- because the conversion alone doubles the host footprint
        fp32 = None
        if not args.no_cpu_fp32:
            sd32 = {k: v.float() for k, v in sd_cpu.items()}
            del sd_cpu
            t32 = cpu_baseline(cfg, sd32, args.regions, args.prompt_len, min(3, g_cpu), dtype=torch.float32)
            del sd32
            tot32 = t32["vit_x2"] + t32["region_proj_splice"] + t32["prefill"] + (G - 1) * t32["decode_step"]
            fp32 = {"value": round(G / tot32, 4), "measured_s": {k: round(v, 4) for k, v in t32.items()}}
        cpu = {"value": round(G / total, 4), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
               "dtype": "bf16 (the reference's eval dtype, eval_spatial.py:221-237)",
               "fp32_value": None if fp32 is None else fp32["value"], "fp32_measured_s": None if fp32 is None else fp32["measured_s"],
               "threads": torch.get_num_threads(), "host_cores": os.cpu_count(),
               "threads_probed": "1 .. 256 on this host class (scripts/cpu_probe.py): 16 fastest for torch's CPU bf16 GEMMs, >64 collapses",
               "port_vs_reference": ("the oracle takes 0.82x (fp32) / 0.98x (bf16) of the REAL reference generate()'s time on the same "
                                     "weights, inputs and threads, ids identical (scaled-down geometry, build container: "
                                     "profiles/r02_cpu_reference_vs_oracle.txt) -- /root/reference does not exist on the bench box"),
               "sample": (f"oracle (CPU restatement of the reference path, bf16, same weights) at FULL depth ({cfg.layers} LLM layers, "
                          f"{cfg.vit_layers_run} ViT layers x2 images) on one request: vision {t['vit_x2']:.2f}s + region/projector/splice "
                          f"{t['region_proj_splice']:.2f}s + prefill(T={spliced_len(cfg, args.prompt_len)}, lm_head on all rows) {t['prefill']:.2f}s "
                          f"measured once, {g_cpu} of the {G} decode steps measured ({t['decode_step'] * 1e3:.0f} ms/token) and scaled "
                          f"to {G - 1}"),
               "measured_s": {k: round(v, 4) for k, v in t.items()}}

    if rank == 0:
        line = {
            "metric": f"region-grounded output tokens/sec @ {'VILA1.5-8B' if args.model.startswith('vila15_8b') else args.model}, {args.regions} regions, greedy",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16" if args.weights == "bf16" else "bf16 activations / fp8-e4m3fn LLM weights (W8A16)" if args.weights == "fp8"
                      else "fp8-e4m3fn LLM weights; prefill W8A8 (per-token e4m3 activations, fp8 MFMA), decode W8A16"), "data": "synthetic (seeded random weights of the named architecture; random images/depth/box masks/ids)",
            "config": {"workload": workload_name(args, cfg, spliced_len(cfg, args.prompt_len)),
                       "requests_per_step_per_gpu": args.batch, "new_tokens_per_request": G, "parallelism": f"dp{world}",
                       "decode": "hipGraph" if not args.no_graph else "eager", "build_s": round(t_build, 1),
                       "llm_weights": args.weights, "dist": dist_info},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist_barrier(device.index)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
