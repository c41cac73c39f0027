"""Turn the raw outputs of scripts/profile_round.sh (gpurun_out/<tag>_*) into the annotated files under profiles/."""
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05_final"
g, p = "gpurun_out/" + tag, "profiles/" + tag
d = json.loads(open(g + "_bench_default.json").read().strip().splitlines()[-1])
open(p + "_bench.json", "w").write(json.dumps(d, indent=1) + "\n")
ks = open(g + "_kernel_stats.txt").read()
m = re.search(r"^\s*\d+\s+[\d.]+\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+gemv_kernel<.*?, true, 2(?:, \d+)?>", ks, re.M)
trace_us = float(m.group(1)) if m else float("nan")
r = d["roofline"]
hdr = f"""# {tag}: rocprofv3 --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline   (scripts/profile_round.sh)
# 1x MI355X, VILA1.5-8B geometry, bs=1, 128 new tokens; 3 generate() calls + the roofline leg in the trace.
# un-profiled default bench of the same build ({p}_bench.json): {d['value']} tok/s, {d['ms_per_step']} ms/request,
#   decode {r['decode_ms_per_token']} ms/token; bench.py's event timing of the gate/up GEMV: {r['avg_launch_ms'] * 1e3:.1f} us/launch
#   (= {r['achieved']} GB/s) vs the trace's {trace_us:.2f} us average below (events include the ~2 us launch gap).
"""
open(p + "_kernel_stats.txt", "w").write(hdr + ks)
open(p + "_prefix.txt", "w").write("# per-request prefix (2 ViT passes as one batch of 2, refinement, pooling, projector, splice, prefill T=259) from the same trace\n"
                                   + open(g + "_prefix.txt").read())
out = [f"# {tag}: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 8   (own pass, no trace domains)",
       "# rocprofv3 sums a counter over the 8 XCDs: wall cycles of a kernel = GRBM_GUI_ACTIVE / 8; SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles per SIMD",
       "# MFMA utilisation = MFMA_BUSY / (wall_cycles * 256 CUs * 4 SIMDs)",
       f"{'dispatches':>10} {'MFMA_BUSY':>16} {'GUI_ACTIVE':>14} {'MFMA util %':>12}  kernel"]
for l in open(g + "_pmc_mfma.txt").read().splitlines():
    mm = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", l)
    if not mm:
        continue
    nd, mf, gui, name = int(mm.group(1)), float(mm.group(2)), float(mm.group(4)), mm.group(5)
    if mf > 0:
        out.append(f"{nd:10d} {mf:16.0f} {gui:14.0f} {100 * mf / ((gui / 8) * 1024):12.1f}  {name[:100]}")
out += ["# the GEMMs of the bs=1 workload are small (M = 259 prefill, M = 1458 ViT: 20-140 us each) and latency-bound; shapes that fill the chip",
        "# go through gemm256 (profiles/r02_gemm256_*.txt: 1.10 PF/s at 4096^3 = 80 % of the same kernel's MFMA-only rate at the sustained clock)."]
open(p + "_pmc_mfma.txt", "w").write("\n".join(out) + "\n")
out = [f"# {tag}: rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 8   (own pass)",
       "# FETCH_SIZE: KiB per dispatch, summed here over dispatches; gfx950 counts a 128-B streaming request as 64 B -> corrected bytes = 2 x (MI355X_MICROARCH.md, HBM section)",
       f"{'dispatches':>10} {'FETCH_SIZE KiB (sum)':>22} {'corrected MiB/dispatch':>24}  kernel"]
gu = None
for l in open(g + "_pmc_fetch.txt").read().splitlines():
    mm = re.match(r"\s*(\d+)\s+([\d.]+)\s+(.*)", l)
    if not mm:
        continue
    nd, fs, name = int(mm.group(1)), float(mm.group(2)), mm.group(3)
    out.append(f"{nd:10d} {fs:22.1f} {2 * fs / 1024 / nd:24.2f}  {name[:100]}")
    if "gemv_kernel" in name and re.search(r", true, 2(, \d+)?>", name):
        gu = fs / nd
if gu:
    out.append(f"# gate/up GEMV (gemv_kernel<..., true, 2>): {2 * gu / 1024:.1f} MiB per dispatch vs 224.0 MiB of weights (2*14336*4096*2 B): ratio {2 * gu * 1024 / (2 * 14336 * 4096 * 2):.4f} -> every weight byte read once.")
    sys.path.insert(0, ".")
    from bench import kernel_source_sha256  # the summary is bound to the kernel source it was measured on (bench.py checks it)

    json.dump({"kernel": "gemv_kernel<bf16,1,swiglu>", "fetch_size_kib": round(gu, 1), "correction": 2.0,
               "traffic_bytes_per_launch": int(round(gu * 2048, -3)), "source": p + "_pmc_fetch.txt",
               "kernel_source_sha256": kernel_source_sha256()}, open("profiles/" + tag.split("_")[0] + "_pmc_gemv.json", "w"))
open(p + "_pmc_fetch.txt", "w").write("\n".join(out) + "\n")
print("composed", p, "| gate/up trace avg", trace_us, "us | bench", d["value"], "tok/s")
import os
for extra in ("_pmc_lds.txt", "_kernel_stats_b4.txt", "_kernel_stats_fp8a8b8.txt"):
    if os.path.exists(g + extra):
        hdr2 = {"_pmc_lds.txt": "# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --max-new-tokens 4 --batch 8 (own pass)\n# conflict share of a kernel = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE\n",
                "_kernel_stats_b4.txt": "# rocprofv3 --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset config2   (BASELINE configs[2] per-GPU shape: 4 requests)\n",
                "_kernel_stats_fp8a8b8.txt": "# rocprofv3 --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --preset config4   (BASELINE configs[4] per-GPU shape: fp8 weights on the fp8 matrix pipe, 8 requests)\n"}[extra]
        open(p + extra, "w").write(hdr2 + open(g + extra).read())
