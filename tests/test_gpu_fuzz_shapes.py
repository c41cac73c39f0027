"""Seeded random SHAPES through the kernel families' checks of tests/test_gpu_kernels.py (the same checkers: fp32 torch / the oracle's
functions on the CPU, the same tolerances) -- the parametrised lists there pin the shapes the dispatch rules were written around; this
draws the ones nobody thought of (ragged N, K tails of every tile size, odd head counts, cache lengths at no boundary), within each
entry point's documented constraints (include/srgpt.h).  SRGPT_FUZZ_CASES=<n> widens the sweep (default 6 draws per family)."""
import os

import pytest
import torch

from tests import test_gpu_kernels as tk

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("SRGPT_FUZZ_CASES", "6"))


def _draw(seed):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    pick = lambda xs: xs[ri(0, len(xs) - 1)]  # noqa: E731
    return ri, pick


@pytest.mark.parametrize("seed", range(N_CASES))
def test_gemm_random_shapes(seed):
    ri, pick = _draw(7000 + seed)
    dtype = pick([torch.bfloat16, torch.bfloat16, torch.bfloat16, torch.float32])
    M = pick([ri(1, 64), ri(65, 400), ri(225, 272), ri(384, 1600), ri(1, 3000)])
    N = pick([ri(1, 300), ri(100, 5000), 8 * ri(1, 1200)])
    K = (8 if dtype == torch.bfloat16 else 4) * pick([ri(1, 40), ri(8, 600), ri(100, 1800)])
    if M * N > 6_000_000:
        N = max(1, 6_000_000 // M)
    tk._gemm_case(dtype, M, N, K)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_gemv_random_shapes(seed):
    ri, pick = _draw(7100 + seed)
    dtype = pick([torch.bfloat16, torch.bfloat16, torch.float32])
    B = pick([1, 1, ri(2, 4), ri(5, 16), ri(17, 33)])
    K = 8 * pick([ri(1, 64), ri(32, 700), ri(256, 1800)])
    while B * K * (2 if dtype == torch.bfloat16 else 4) > 140 * 1024 and B <= 4:  # the VALU kernel stages all rows in LDS
        K -= 512
    N = pick([ri(1, 200), ri(100, 9000), ri(1000, 40000)])
    tk.test_gemv_variants(dtype, B, N, K)
    tk.test_gemv_swiglu(dtype, B, max(1, N // 2), K)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_gemv_w8_random_shapes(seed):
    ri, pick = _draw(7200 + seed)
    B = pick([1, ri(2, 4), ri(5, 8), ri(9, 16), ri(17, 24)])
    K = 8 * pick([ri(2, 64), ri(32, 700), ri(256, 1800)])
    N = pick([ri(1, 200), ri(100, 9000), ri(1000, 33000)])
    tk.test_gemv_w8_variants(B, N, K)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_attention_random_shapes(seed):
    ri, pick = _draw(7300 + seed)
    dtype = pick([torch.bfloat16, torch.bfloat16, torch.float32])
    Hkv = ri(1, 4)
    Hq = Hkv * pick([1, 1, 2, 3, 4, 8])
    D = pick([16, 32, 64, 72, 80, 96, 128])
    causal = bool(ri(0, 1))
    Tk = pick([ri(1, 70), ri(60, 800), ri(700, 2300)])
    Tq = Tk if (not causal or ri(0, 2)) else ri(1, Tk)
    B = ri(1, 3) if Tk < 900 else 1
    tk.test_attention(dtype, B, Tq, Tk, Hq, Hkv, D, causal)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_decode_attention_random_shapes(seed):
    ri, pick = _draw(7400 + seed)
    dtype = pick([torch.bfloat16, torch.bfloat16, torch.float32])
    Hkv = ri(1, 8)
    Hq = Hkv * pick([1, 2, 4, 8])
    D = pick([16, 32, 64, 128, 128, 128])
    max_pos = pick([512, 1024, 2048, 4096])
    P = pick([ri(0, 70), ri(0, max_pos - 1), ri(max_pos // 2, max_pos - 1)])
    B = ri(1, 4) if P < 1500 else 1
    tk.test_rope_append_and_decode_attention(dtype, B, Hq, Hkv, D, P, max_pos)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_norms_random_shapes(seed):
    ri, pick = _draw(7500 + seed)
    dtype = pick([torch.bfloat16, torch.float32])
    cols = (8 if dtype == torch.bfloat16 else 4) * pick([ri(1, 40), ri(30, 700), ri(500, 2000)])
    tk.test_layernorm_rmsnorm(dtype, ri(1, 300), cols)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_fused_gemm_random_shapes(seed):
    """the one-call forms against their two launches, BIT for bit, wherever the dispatch sends the shape"""
    ri, pick = _draw(7600 + seed)
    dtype = pick([torch.bfloat16, torch.bfloat16, torch.bfloat16, torch.float32])
    M = pick([ri(1, 64), ri(65, 400), ri(225, 272), ri(384, 1500)])
    K = 8 * pick([ri(1, 40), ri(8, 600), ri(100, 1800)])
    N = 8 * pick([ri(1, 64), ri(32, 1100), ri(512, 2100)])
    while M * N > 4_000_000:
        N //= 2
    N = max(8, N // 8 * 8)
    tk.test_gemm_norm_equals_the_two_launches(dtype, M, N, K, bool(ri(0, 1)))
    tk.test_gemm_swiglu_equals_the_two_launches(dtype, M, max(8, N // 16 * 8), K)  # srgpt_silu_mul: rows of whole 16-byte chunks
    Hkv = ri(1, 4)
    Hq, D = Hkv * pick([1, 2, 4, 8]), pick([32, 64, 128])
    B = ri(1, 3)
    T = pick([ri(1, 40), ri(30, 300), ri(225, 272)])
    tk.test_gemm_rope_kv_append_equals_the_two_launches(dtype, B, T, Hq, Hkv, D, min(K, 4096), bool(ri(0, 1)))
    if dtype == torch.bfloat16:
        tk.test_gemm_w8_equals_gemm_on_dequantised_weights(M, N, K)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_row_statistics_and_packed_layout_random_shapes(seed):
    ri, pick = _draw(7700 + seed)
    fp8 = bool(ri(0, 1))
    B = pick([ri(2, 4), ri(5, 8), ri(9, 16), ri(17, 20)])
    N = 64 * pick([ri(1, 16), ri(8, 100)])        # the consumer's K: whole 64-k blocks (fp8) / 32-k blocks (bf16)
    K = 64 * pick([ri(1, 16), ri(8, 230)])
    rows = pick([4, 8, 16])
    N2 = 2 * rows * pick([ri(1, 20), ri(10, 500)])
    tk.test_gemv_rowss_handoff(fp8, B, N, K, pick([N2, ri(1, 3000)]))
    tk.test_gemv_rowss_packed_equals_row_major_bit_for_bit(fp8, B, N, K, N2, rows)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_region_pool_and_loss_random_shapes(seed):
    ri, pick = _draw(7800 + seed)
    dtype = pick([torch.bfloat16, torch.float32])
    fw = pick([24, 27, 48, 64, 96, 108])
    S = fw * pick([1, 2, 3, 4]) if ri(0, 1) else pick([96, 192, 224, 336, 384, 448])
    tk.test_region_pool_true_shape_vs_oracle(dtype, ri(1, 20), S, fw, 8 * ri(1, 160))
    tk.test_cross_entropy_vs_torch(ri(1, 80), pick([ri(2, 300), ri(300, 40000), 128258]))


@pytest.mark.parametrize("seed", range(N_CASES))
def test_decode_step_composition_and_invariants_random(seed):
    """the graph-captured batched step == the per-entry composition at ANY row count (17 = 16 + a single-row tail included); decode
    attention bits independent of the cache capacity; the beam permutation for every beam count the kernel serves"""
    ri, pick = _draw(7900 + seed)
    tk.test_batched_decode_step_equals_rowss_composition_bit_for_bit(pick([ri(2, 16), ri(2, 16), 17, ri(18, 24)]), bool(ri(0, 1)))
    Hkv = pick([1, 2, 4, 8, 20])
    tk.test_decode_attention_bits_do_not_depend_on_cache_capacity(ri(1, 3), Hkv * pick([1, 2, 4, 8] if Hkv < 20 else [1]), Hkv, ri(1, 760))  # the smallest cache of that test holds 768 positions
    tk.test_kv_beam_reorder_equals_index_select(ri(2, 8), ri(1, 3), pick([torch.bfloat16, torch.float32]))


@pytest.mark.parametrize("seed", range(N_CASES))
def test_fp8_matrix_pipe_random_shapes(seed):
    """W8A8 prefill pieces: the fused quantisers against their two launches (bit for bit), the fp8 GEMM against fp64 on the dequantised
    operands (K in whole 128-k tiles, at least two)"""
    from tests import test_gpu_fp8_mfma as tf
    ri, pick = _draw(8000 + seed)
    M = pick([ri(1, 64), ri(65, 400), ri(225, 300), ri(384, 2100)])
    tf.test_quant_rows_fused_rmsnorm_is_the_two_launches_bit_for_bit(max(2, min(M, 300)), 8 * pick([ri(1, 64), ri(32, 700), ri(256, 2048)]))
    tf.test_quant_rows_fused_swiglu_is_the_two_launches_bit_for_bit(max(2, min(M, 300)), 8 * pick([ri(1, 64), ri(32, 700), ri(256, 2048)]))
    K = 128 * pick([ri(2, 8), ri(4, 40), ri(16, 112)])
    N = pick([ri(1, 300), ri(100, 3000), 8 * ri(1, 800)])
    while M * N * K > 2_500_000_000:  # the fp64 reference runs on the host
        N = max(1, N // 2)
    tf.test_gemm_w8a8_equals_gemm_of_dequantised_operands(M, N, K)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_sampler_kept_set_random_settings(seed):
    from tests import test_gpu_sampling as ts
    ri, pick = _draw(8100 + seed)
    temperature = pick([0.05, 0.2, 0.7, 1.0, 1.3, 2.5]) * (1 + 0.01 * ri(0, 30))
    top_p = pick([None, None, 1.0, 0.01 * ri(2, 99)])
    ts.test_kept_set_equals_hf_warpers(temperature, ri(1, 64), top_p, pick([ri(70, 2000), ri(2000, 40000), ri(40000, 160000)]))


@pytest.mark.parametrize("seed", range(N_CASES))
def test_preprocessing_random_sizes(seed):
    """process_images_device / process_regions_device on raw uint8 of ANY size == the host path (PIL bicubic + rescale / normalise; the
    cv2-nearest restatement; the pad mode's soft-edged square) bit for bit -- up- and down-sampling, extreme aspect ratios, tiny inputs"""
    import numpy as np
    from PIL import Image
    from types import SimpleNamespace

    from spatialrgpt_amd.mm_utils import (SrgptImageProcessor, process_images, process_images_device, process_regions,
                                          process_regions_device)

    ri, pick = _draw(8200 + seed)
    rng = np.random.default_rng(8200 + seed)
    size = pick([384, 384, 336, 378, 224])
    proc = SrgptImageProcessor(size=size)
    dim = lambda: pick([ri(2, 40), ri(30, 500), ri(300, 1300)])  # noqa: E731
    shapes = [(dim(), dim()) for _ in range(3)]
    for mode in ("resize", "pad", None):
        cfg = SimpleNamespace(image_aspect_ratio=mode, image_processor=proc)
        ims = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in shapes]
        ref = process_images(ims, proc, cfg)
        got = process_images_device(ims, proc, cfg, device="cuda", dtype=torch.float32)
        assert got.shape == ref.shape and torch.equal(got.cpu(), ref), (mode, shapes, size, float((got.cpu() - ref).abs().max()))
    for mode in ("resize", "pad"):
        cfg = SimpleNamespace(image_aspect_ratio=mode, image_processor=proc)
        h, w = dim(), dim()
        hi = pick([1, 255])
        mk = [(rng.random((h, w)) > 0.6).astype(np.uint8) * hi for _ in range(ri(1, 4))]
        mk[0][h // 4:h // 2 + 1, w // 4:w // 2 + 1] = hi
        refm = process_regions(mk, proc, cfg)
        gotm = process_regions_device(mk, proc, cfg, device="cuda", dtype=torch.float32)
        assert gotm.shape == refm.shape and torch.equal(gotm.cpu(), refm), (mode, (h, w), size, float((gotm.cpu() - refm).abs().max()))
