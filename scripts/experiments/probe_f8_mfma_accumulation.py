"""How v_mfma_scale_f32_16x16x128_f8f6f4 adds its 128 products, seen through srgpt_gemm_w8a8 (whose epilogue rounds to bf16, so
both probes are built to have bf16-exact answers).
 1. flat data: every product in [1, 3.52]; the output must be the correctly rounded exact sum.
 2. cancellation: +P and -P (P = 2^16) next to 126 equal small products s in the same 128-wide block, P / s = 2^d, d = 0..28.
    The exact answer is 126 s (6 significant bits).  If the adder aligned the products to the largest one and dropped what falls
    below its window, the small terms would vanish for large d; an fp32 accumulation of exact products keeps them for d <= 23 - 7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spatialrgpt_amd import ops

dev = "cuda"


def code(x):  # e4m3fn byte of an exactly representable value
    return torch.tensor([x], dtype=torch.float32).to(torch.float8_e4m3fn).view(torch.uint8).item()


g = torch.Generator().manual_seed(1)
M = N = 256
K = 512
a8 = (torch.randint(0x38, 0x40, (M, K), generator=g) | (torch.randint(0, 2, (M, K), generator=g) << 7)).to(torch.uint8)
w8 = (torch.randint(0x38, 0x40, (N, K), generator=g) | (torch.randint(0, 2, (N, K), generator=g) << 7)).to(torch.uint8)
one = torch.ones((M,), dtype=torch.float32)
ref = a8.view(torch.float8_e4m3fn).double() @ w8.view(torch.float8_e4m3fn).double().T
out = ops.gemm_w8a8(a8.to(dev), one.to(dev), w8.to(dev), one.to(dev), out_f32=True).double().cpu()
print(f"flat data {M}x{N}x{K}: elements that are not the correctly rounded exact sum: "
      f"{int((out != ref.to(torch.bfloat16).double()).sum())} / {ref.numel()}")

K = 256
a8 = torch.zeros((M, K), dtype=torch.uint8)
w8 = torch.zeros((N, K), dtype=torch.uint8)
for m in range(15):
    a8[m, 0] = code(2.0 ** 8)
    a8[m, 1] = code(-(2.0 ** 8))
    a8[m, 2:128] = code(2.0 ** (-6 + m))       # P_a / s_a = 2^(14 - m)
    w8[m, 0] = code(2.0 ** 8)
    w8[m, 1] = code(2.0 ** 8)
    w8[m, 2:128] = code(2.0 ** (-6 + m))
out = ops.gemm_w8a8(a8.to(dev), one.to(dev), w8.to(dev), one.to(dev), out_f32=True).double().cpu()
print("+P - P + 126 s in one 128-wide block, P = 2^16, d = log2(P / s): result / (126 s)")
seen = {}
for m in range(15):
    for n in range(15):
        seen.setdefault(28 - m - n, []).append(float(out[m, n]) / (126 * 2.0 ** (-12 + m + n)))
for d in sorted(seen):
    print(f"  d = {d:2d}: " + " ".join(f"{v:.4f}" for v in seen[d][:4]))
