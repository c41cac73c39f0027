"""GPU parity tests of the opt-in W8A8 prefill (BASELINE configs[4] "fp8 weights on CDNA4 fp8 MFMA"): per-token e4m3
activation quantisation (srgpt_quant_rows_e4m3) and the fp8 x fp8 GEMM on the fp8 matrix pipe (srgpt_gemm_w8a8), through the
C ABI.  Checkers: the host quantisation rule (bit-exact), the fp64 GEMM of the dequantised operands (fp32-accumulation
tolerance, stated below), and the oracle's restatement of the mode (llama_forward(act_quant=fp8_rowwise_fake_quant))."""
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from spatialrgpt_amd import _lib, ops
    return ops, _lib


def _deq(q8, sc):
    return q8.view(torch.float8_e4m3fn).double() * sc.double()[:, None]


def _acts(M, K, seed, dtype=torch.bfloat16):
    """activations with per-row magnitudes over several octaves and a few outlier channels"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((M, K), generator=g) * torch.exp2(torch.randint(-6, 5, (M, 1), generator=g).float())
    x[:, torch.randint(0, K, (4,), generator=g)] *= 40.0
    return x.to(dtype)


def test_quant_rows_e4m3_equals_host_rule():
    """codes and scales are bit-identical to spatialrgpt_amd.ops.quantize_fp8_rows (frexp / ldexp power-of-two scale,
    round-to-nearest-even e4m3fn) evaluated by torch on the CPU; rows of zeros, a row whose maximum is exactly 448 * 2^k, a
    strided input."""
    ops, _ = _ops()
    x = _acts(77, 640, 11)
    x[5] = 0
    x[6] = 0
    x[6, 17] = 448.0 * 4
    x[7, :] = torch.linspace(-3.0, 3.0, 640).to(torch.bfloat16)  # dense in the low binades: subnormal codes after scaling by the max
    x[7, 0] = 3000.0
    q_ref, sc_ref, _ = ops.quantize_fp8_rows(x)
    q, sc = ops.quant_rows_e4m3(x.to(DEV))
    assert torch.equal(sc.cpu(), sc_ref), "row scales"
    bad = int((q.cpu() != q_ref).sum())
    assert bad == 0, f"{bad}/{q_ref.numel()} e4m3 codes differ from the host rule"
    wide = torch.zeros((77, 768), dtype=torch.bfloat16)
    wide[:, :640] = x
    wide[:, 640:] = 1e4  # must not be read
    q2, sc2 = ops.quant_rows_e4m3(wide.to(DEV)[:, :640])
    assert torch.equal(q2.cpu(), q_ref) and torch.equal(sc2.cpu(), sc_ref), "strided rows"


@pytest.mark.parametrize("M,K", [(5, 64), (77, 640), (259, 4096), (33, 2560), (3, 16384)])
def test_quant_rows_fused_rmsnorm_is_the_two_launches_bit_for_bit(M, K):
    """srgpt_quant_rows_e4m3_rmsnorm (what the W8A8 prefill runs) == srgpt_rmsnorm then srgpt_quant_rows_e4m3: codes and
    scales identical, incl. a row of zeros, an outlier column, a strided input, and one to eight chunks per thread."""
    ops, _ = _ops()
    x = _acts(M, K, 21).to(DEV)
    x[0] = 0
    g = (1.0 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16).to(DEV)
    q_ref, sc_ref = ops.quant_rows_e4m3(ops.rmsnorm(x, g, 1e-5))
    q, sc = ops.quant_rows_e4m3_rmsnorm(x, g, 1e-5)
    assert torch.equal(sc, sc_ref), "row scales"
    assert torch.equal(q, q_ref), f"{int((q != q_ref).sum())}/{q.numel()} codes differ"
    wide = torch.full((M, K + 64), 1e4, device=DEV, dtype=torch.bfloat16)
    wide[:, :K] = x
    q2, sc2 = ops.quant_rows_e4m3_rmsnorm(wide[:, :K], g, 1e-5)
    assert torch.equal(q2, q_ref) and torch.equal(sc2, sc_ref), "strided rows"


@pytest.mark.parametrize("M,inter", [(5, 48), (77, 640), (259, 14336), (33, 11008), (9, 6912)])
def test_quant_rows_fused_swiglu_is_the_two_launches_bit_for_bit(M, inter):
    """srgpt_quant_rows_e4m3_swiglu == srgpt_silu_mul then srgpt_quant_rows_e4m3 on [gate | up] rows."""
    ops, _ = _ops()
    gu = _acts(M, 2 * inter, 22).to(DEV)
    gu[1] = 0
    q_ref, sc_ref = ops.quant_rows_e4m3(ops.silu_mul(gu))
    q, sc = ops.quant_rows_e4m3_swiglu(gu)
    assert torch.equal(sc, sc_ref), "row scales"
    assert torch.equal(q, q_ref), f"{int((q != q_ref).sum())}/{q.numel()} codes differ"


def test_quant_rows_fused_rejects_rows_wider_than_its_register_tile():
    ops, _ = _ops()
    with pytest.raises(NotImplementedError):
        ops.quant_rows_e4m3_swiglu(torch.zeros((2, 2 * 16392), device=DEV, dtype=torch.bfloat16))

@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (77, 333, 384), (259, 6144, 4096), (300, 520, 1408), (2072, 1024, 512),
                                   (513, 4096, 14336)])
def test_gemm_w8a8_equals_gemm_of_dequantised_operands(M, N, K):
    """Against the fp64 GEMM of the dequantised operands.  Every product is exact and the sum is kept in fp32, so the result
    before the epilogue differs from the reference only by the order of the fp32 additions and the adder's alignment window (bounded here by 2e-4 of
    sum_k |a_k w_k|; scripts/experiments/probe_f8_mfma_accumulation.py looks at the adder itself); the epilogue then rounds to bf16 at
    srgpt_gemm's points (the Linear's output; the residual sum), also when the store is fp32.  Checked: every element within
    half a bf16 ulp + the accumulation bound of the reference, and (printed, bounded) the share of elements that ARE the
    correctly rounded reference -- a swapped fragment, a wrong row or a wrong scale is O(1) on both.  Ragged M / N tiles, the
    split-K slabs (small tile grids), bias + residual."""
    ops, L = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a8, asc, _ = ops.quantize_fp8_rows(_acts(M, K, 1))
    w8, wsc, _ = ops.quantize_fp8_rows((torch.randn((N, K), generator=g) * 0.03).to(torch.bfloat16))
    wsc = wsc * torch.exp2(torch.randint(-3, 4, (N,), generator=g).float())  # make the column scales asymmetric
    A, W = _deq(a8, asc), _deq(w8, wsc)
    ref = A @ W.T
    acc_tol = 2e-4 * (A.abs() @ W.abs().T)
    out = ops.gemm_w8a8(a8.to(DEV), asc.to(DEV), w8.to(DEV), wsc.to(DEV), out_f32=True)
    assert out.dtype == torch.float32
    ref_r = ref.to(torch.bfloat16).double()
    err = (out.double().cpu() - ref).abs()
    lim = 2.0 ** -8 * (ref.abs() + acc_tol) + acc_tol + 1e-30
    same = float((out.double().cpu() == ref_r).double().mean())
    print(f"\ngemm_w8a8 {M}x{N}x{K}: max err / limit = {float((err / lim).max()):.3f}, equal to the rounded reference: {same:.5f}")
    assert float((err / lim).max()) <= 1.0, "plain output outside half a bf16 ulp + the accumulation bound"
    assert same > 0.95, f"only {same:.4f} of the elements equal the correctly rounded reference"
    # bias and residual: round(acc * scales + bias) -> round(+ residual), as srgpt_gemm materialises them
    b = (torch.randn((N,), generator=g) * 0.1).to(torch.bfloat16)
    r = (torch.randn((M, N), generator=g)).to(torch.bfloat16)
    mid = (ref + b.double()).to(torch.bfloat16).double()
    ref2 = (mid + r.double()).to(torch.bfloat16)
    out2 = ops.gemm_w8a8(a8.to(DEV), asc.to(DEV), w8.to(DEV), wsc.to(DEV), bias=b.to(DEV), residual=r.to(DEV))
    assert out2.dtype == torch.bfloat16
    d = (out2.double().cpu() - ref2.double()).abs()
    lim2 = 2.0 ** -7 * (mid.abs() + ref2.double().abs()) + 2 * acc_tol + 1e-30  # one ulp at either rounding point
    same2 = float((out2.cpu() == ref2).double().mean())
    print(f"  + bias + residual: max err / limit = {float((d / lim2).max()):.3f}, equal to the rounded reference: {same2:.5f}")
    assert float((d / lim2).max()) <= 1.0, "bias + residual output outside the rounding limits"
    assert same2 > 0.95, f"only {same2:.4f} of the elements equal the correctly rounded reference"


def test_gemm_w8a8_rejects_unsupported_shapes():
    ops, L = _ops()
    a8 = torch.zeros((16, 192), dtype=torch.uint8, device=DEV)
    w8 = torch.zeros((16, 192), dtype=torch.uint8, device=DEV)
    sc = torch.ones((16,), dtype=torch.float32, device=DEV)
    with pytest.raises(Exception, match="multiple of 128"):
        ops.gemm_w8a8(a8, sc, w8, sc)


@pytest.mark.parametrize("batch", [1, 3])
def test_w8a8_prefill_vs_oracle_restatement(batch):
    """llm_weight_format="fp8_w8a8": all-position prefill logits against the oracle run on the dequantised weights WITH the
    per-token activation quantisation in front of the seven layer projections (the mode's restatement), at the 3e-2-of-max
    bar of the W8A16 test; and what the mode costs against W8A16 (same weights, bf16 activations), reported and bounded.
    Decode after a W8A8 prefill is W8A16 on the cache the prefill wrote: teacher-forced against the oracle doing the same."""
    from oracle import srgpt_oracle as so
    from spatialrgpt_amd.config import SrgptConfig
    from spatialrgpt_amd.engine import SrgptEngine
    import ctypes as C
    from spatialrgpt_amd import _lib as L, ops

    dtype = torch.bfloat16
    kw = dict(vit_hidden=64, vit_inter=128, vit_layers=2, vit_heads=4, image_size=56, patch_size=14, hidden=512, inter=1408,
              layers=3, heads=8, kv_heads=2, vocab=1000, mask_token_id=998, depth_token_id=999, rope_theta=10000.0)
    ocfg = so.SrgptConfig(**kw)
    w = so.synth_weights(ocfg, seed=3, dtype=dtype)
    wq = so.fp8_dequantised_weights(w)
    eng = SrgptEngine(SrgptConfig(**kw), dict(w), device=DEV, dtype=dtype, rope_positions=512, llm_weight_format="fp8_w8a8")
    assert eng.w.fp8_act and eng.w.llm.fp8_act == 1 and eng.w.llm_weight_format == "fp8"
    g = torch.Generator().manual_seed(5)
    T, G = 37, 3
    x = (torch.randn((batch, T, 512), generator=g) * 0.5).to(dtype)
    st, all_logits, _ = eng.prefill(x.to(DEV), max_new=G + 1, all_logits=True)
    pos = torch.arange(T)[None].expand(batch, -1)
    kv = so.KVCache(ocfg.layers)
    ref = so.llama_forward(wq, ocfg, x, pos, kv, act_quant=so.fp8_rowwise_fake_quant)
    tol = 3e-2 * float(ref.abs().max())
    assert_close(all_logits, ref, tol, 0, "W8A8 prefill logits vs the oracle's restatement")
    ref16 = so.llama_forward(wq, ocfg, x, pos, so.KVCache(ocfg.layers))
    dev = float((all_logits.float().cpu() - ref16).pow(2).mean().sqrt() / ref16.pow(2).mean().sqrt())
    print(f"\nW8A8 vs W8A16 prefill logits: relative rms deviation {dev:.4f} (batch {batch})")
    assert dev < 0.15, f"W8A8 prefill deviates {dev:.3f} rms from W8A16"
    # decode continues in W8A16 on the cache the W8A8 prefill wrote: teacher-forced against the oracle doing the same
    lib = L.load()
    L.check(lib.srgpt_llm_sample_first(C.byref(eng.w.llm), C.byref(st.c), ops._stream()))
    for s in range(G):
        L.check(lib.srgpt_llm_decode_step(C.byref(eng.w.llm), C.byref(st.c), ops._stream()))
        tok = st.out_ids[:, s].cpu()
        emb = torch.nn.functional.embedding(tok[:, None], wq["llm.model.embed_tokens.weight"])
        r = so.llama_forward(wq, ocfg, emb, torch.full((batch, 1), T + s), kv)
        assert_close(st.logits, r[:, 0], tol, 0, f"W8A16 decode step {s} on the W8A8-written cache vs the oracle")
