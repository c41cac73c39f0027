"""fp8 x fp8 GEMM (srgpt_gemm_w8a8, fp8 matrix pipe) against srgpt_gemm_w8 (W8A16, bf16 matrix pipe) and the per-token
quantisation pass at the LLM prefill shapes of BASELINE configs[4] (8 requests: M = 2072) and of one request (M = 259), plus a
square shape.  Graph-captured, distinct weights per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import _lib
if os.environ.get("SRGPT_LIB"):  # a variant build
    _lib.LIB_PATH = os.path.abspath(os.environ["SRGPT_LIB"])
from spatialrgpt_amd import ops

dev = "cuda"
shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate/up", 28672, 4096), ("down", 4096, 14336)]
cases = [(f"{n} x8", 2072, N, K) for n, N, K in shapes] + [(f"{n} x1", 259, N, K) for n, N, K in shapes] + [("sq 4096", 4096, 4096, 4096),
                                                                                                          ("sq 8192", 8192, 8192, 8192)]
side = torch.cuda.Stream()


def timed(fn, L):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * L)


def quick():
    """--quick: output check against torch's fp32 GEMM of the dequantised operands, a K sweep at 4096 x 4096 (one tile per CU:
    time = fixed cost + K tiles x slope) and the 8-request shapes, W8A8 only."""
    g = torch.Generator(device=dev).manual_seed(3)
    for M, N, K in [(300, 520, 1408), (2072, 6144, 4096)]:
        a = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn((N, K), device=dev, generator=g) * 0.03).to(torch.bfloat16)
        r = torch.randn((M, N), device=dev, generator=g).to(torch.bfloat16)
        a8, asc = ops.quant_rows_e4m3(a)
        w8, wsc, _ = ops.quantize_fp8_rows(w)
        ref = (a8.view(torch.float8_e4m3fn).float() * asc[:, None]) @ (w8.view(torch.float8_e4m3fn).float() * wsc[:, None]).T
        o1 = ops.gemm_w8a8(a8, asc, w8, wsc)
        o2 = ops.gemm_w8a8(a8, asc, w8, wsc, residual=r, out_f32=True)
        r1 = ref.to(torch.bfloat16)
        r2 = (r1.float() + r.float()).to(torch.bfloat16).float()
        print(f"check {M}x{N}x{K}: equal to the rounded fp32 reference: plain {float((o1 == r1).float().mean()):.4f} "
              f"(max rel err {float(((o1.float() - ref).abs() / (ref.abs() + 1e-2)).max()):.3e}), + residual, fp32 store "
              f"{float((o2 == r2).float().mean()):.4f}", flush=True)
    for name, M, N, K in [(f"K={k}", 4096, 4096, k) for k in (512, 2048, 8192)] + cases[:4]:
        L = 4 if N * K < 2e8 else 2
        W8 = [torch.randint(0, 120, (N, K), device=dev, dtype=torch.uint8) for _ in range(L)]
        wsc = torch.full((N,), 2.0 ** -9, device=dev)
        a8, asc = ops.quant_rows_e4m3(torch.randn((M, K), device=dev, dtype=torch.bfloat16))
        out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        t8 = timed(lambda: [ops.gemm_w8a8(a8, asc, w, wsc, out=out) for w in W8], L)
        print(f"{name:11s} M={M:5d} N={N:6d} K={K:6d}  W8A8 {t8:8.1f} us {2.0 * M * N * K / t8 / 1e6:7.1f} TF/s", flush=True)


if "--pmc" in sys.argv:  # under rocprofv3 --pmc: a few plain launches of both kernels at the 8-request shapes and 8192^3
    for name, M, N, K in cases[:4] + cases[-1:]:
        w8 = torch.randint(0, 120, (N, K), device=dev, dtype=torch.uint8)
        wsc = torch.full((N,), 2.0 ** -9, device=dev)
        a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
        a8, asc = ops.quant_rows_e4m3(a)
        for _ in range(3):
            ops.gemm_w8a8(a8, asc, w8, wsc)
            ops.gemm_w8(a, w8, wsc)
    torch.cuda.synchronize()
    sys.exit(0)

if "--quick" in sys.argv:
    quick()
    sys.exit(0)

for name, M, N, K in cases:
    L = 4 if N * K < 2e8 else 2
    W8 = [torch.randint(0, 120, (N, K), device=dev, dtype=torch.uint8) for _ in range(L)]
    wsc = torch.full((N,), 2.0 ** -9, device=dev)
    a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    a8, asc = ops.quant_rows_e4m3(a)
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    t16 = timed(lambda: [ops.gemm_w8(a, w, wsc, out=out) for w in W8], L)
    t8 = timed(lambda: [ops.gemm_w8a8(a8, asc, w, wsc, out=out) for w in W8], L)
    tq = timed(lambda: [ops.quant_rows_e4m3(a) for _ in range(L)], L)
    fl = 2.0 * M * N * K
    print(f"{name:11s} M={M:5d} N={N:6d} K={K:6d}  W8A16 {t16:8.1f} us {fl / t16 / 1e6:7.1f} TF/s | W8A8 {t8:8.1f} us {fl / t8 / 1e6:7.1f} TF/s"
          f" | quant {tq:6.1f} us | (W8A8 + quant) / W8A16 = {(t8 + tq) / t16:.2f}", flush=True)
    del W8
