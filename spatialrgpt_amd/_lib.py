"""ctypes binding of libsrgpt_hip.so (the C ABI declared in include/srgpt.h).

The product path has NO fallback: if the HIP extension is missing or fails to load, importing any
compute entry point raises immediately (build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C spatialrgpt_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsrgpt_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_SILU, ACT_QUICK_GELU = 0, 1, 2, 3, 4
OUT_PLAIN, OUT_DECONV2X = 0, 1
ERR_ARG, ERR_UNSUPPORTED, ERR_LAUNCH, ERR_STATE = -1, -2, -3, -4

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class VitWeights(C.Structure):
    _fields_ = [
        ("dtype", i32), ("hidden", i32), ("inter", i32), ("heads", i32), ("n_layers_run", i32),
        ("image_size", i32), ("patch", i32), ("kp", i32), ("inter_pad", i32), ("act", i32), ("eps", f32),
        ("patch_w", vp), ("patch_b", vp), ("pos_emb", vp), ("cls_emb", vp), ("pre_ln_w", vp), ("pre_ln_b", vp),
        ("ln1_w", C.POINTER(vp)), ("ln1_b", C.POINTER(vp)),
        ("wqkv", C.POINTER(vp)), ("bqkv", C.POINTER(vp)),
        ("wo", C.POINTER(vp)), ("bo", C.POINTER(vp)),
        ("ln2_w", C.POINTER(vp)), ("ln2_b", C.POINTER(vp)),
        ("w1", C.POINTER(vp)), ("b1", C.POINTER(vp)),
        ("w2", C.POINTER(vp)), ("b2", C.POINTER(vp)),
    ]


class LlmWeights(C.Structure):
    _fields_ = [
        ("dtype", i32), ("hidden", i32), ("inter", i32), ("layers", i32), ("heads", i32), ("kv_heads", i32),
        ("head_dim", i32), ("vocab", i32), ("rms_eps", f32),
        ("rope_cos", vp), ("rope_sin", vp), ("embed", vp), ("final_norm", vp), ("lm_head", vp),
        ("attn_norm", C.POINTER(vp)), ("wqkv", C.POINTER(vp)), ("wo", C.POINTER(vp)),
        ("mlp_norm", C.POINTER(vp)), ("wgu", C.POINTER(vp)), ("wdown", C.POINTER(vp)),
        ("lm_head8", vp), ("lm_head_scale", vp),
        ("wqkv8", C.POINTER(vp)), ("wqkv_scale", C.POINTER(vp)), ("wo8", C.POINTER(vp)), ("wo_scale", C.POINTER(vp)),
        ("wgu8", C.POINTER(vp)), ("wgu_scale", C.POINTER(vp)), ("wdown8", C.POINTER(vp)), ("wdown_scale", C.POINTER(vp)),
        ("fp8_act", i32),
        ("wqkv8p", C.POINTER(vp)), ("wo8p", C.POINTER(vp)), ("wgu8p", C.POINTER(vp)), ("wdown8p", C.POINTER(vp)),
        ("pk_rows_qkv", i32), ("pk_rows_o", i32), ("pk_rows_gu", i32), ("pk_rows_down", i32),
    ]


class LlmState(C.Structure):
    _fields_ = [
        ("batch", i32), ("max_pos", i32), ("kcache", vp), ("vcache", vp), ("pos", vp), ("tok", vp),
        ("out_ids", vp), ("step", vp), ("max_new", i32), ("ws_tokens", i32), ("ws", vp), ("logits", vp),
        ("sampling", vp),  # ABI 6: device pointer to a Sampling block, or NULL = greedy
    ]


class Sampling(C.Structure):
    """srgpt_sampling (include/srgpt.h): the parameter block of the device-side draw; lives in DEVICE memory (40 bytes)."""
    _fields_ = [("temperature", f32), ("top_k", i32), ("top_p", f32), ("top_p_rm", f32), ("seed", C.c_uint64),
                ("counter", C.c_uint64), ("kept_out", vp)]


SAMPLING_TOP_K_MAX = 64        # sample.hip SMP_K
SAMPLING_KEPT_MAX = 256        # sample.hip SMP_LIST
SAMPLING_VOCAB_MAX = 128 * 2048

NORM_RMS, NORM_LAYER = 1, 2  # srgpt_gemm_norm
SPLICE_STATS = 8             # SRGPT_SPLICE_STATS: ints per prompt the splice plan reports
ROWSS_STRIDE = 512           # SRGPT_ROWSS_STRIDE: slots per row of a row-statistics table
ABI_VERSION = 9  # include/srgpt.h; bumped with every export / layout change

_SIGNATURES = {
    "srgpt_last_error": (C.c_char_p, []),
    "srgpt_abi_version": (i32, []),
    "srgpt_device_cus": (i32, []),
    "srgpt_gemm_ws_bytes": (i64, [i32, i32]),
    "srgpt_gemm": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp]),
    "srgpt_gemm_norm": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, i64, i32, vp, vp, vp, f32, i32, vp]),
    "srgpt_gemm_rope_kv_append": (i32, [vp, vp, vp, i32, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "srgpt_gemm_swiglu": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, i64, i32, vp]),
    "srgpt_gemm_w8": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp]),
    "srgpt_quant_rows_e4m3": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "srgpt_quant_rows_e4m3_rmsnorm": (i32, [vp, vp, f32, vp, vp, i32, i32, i32, vp]),
    "srgpt_quant_rows_e4m3_swiglu": (i32, [vp, vp, vp, i32, i32, vp]),
    "srgpt_gemm_w8a8": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, vp]),
    "srgpt_gemv": (i32, [vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "srgpt_gemv_w8": (i32, [vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "srgpt_splice_scratch_ints": (i64, [i32, i32]),
    "srgpt_splice_plan": (i32, [vp, vp, i32, i32, i32, i32, vp, i32, i32, i64, i64, i32, i32, vp, vp, vp, vp]),
    "srgpt_splice_gather": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp]),
    "srgpt_gemv_rowss_supported": (i32, [i32, i32, i32]),
    "srgpt_gemv_rowss": (i32, [vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]),
    "srgpt_packed_bytes": (C.c_size_t, [i32, i32, i32, i32]),
    "srgpt_pack_decode_weights": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "srgpt_layernorm": (i32, [vp, vp, vp, vp, i32, i32, f32, i32, i32, vp]),
    "srgpt_rmsnorm": (i32, [vp, vp, vp, i32, i32, f32, i32, vp]),
    "srgpt_attention": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, i64, i64, i64, i64,
                              i64, f32, i32, vp, i32, vp]),
    "srgpt_rope_kv_append": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "srgpt_decode_attn_ws_floats": (i64, [i32, i32, i32]),
    "srgpt_decode_attention": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "srgpt_region_pool_ws_floats": (i64, [i32, i32, i32]),
    "srgpt_region_pool": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, i32, i32, vp]),
    "srgpt_region_pool_u8": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, i32, vp]),
    "srgpt_avgpool": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "srgpt_s2d": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "srgpt_im2col": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "srgpt_embed_rows": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "srgpt_kv_beam_reorder": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "srgpt_scatter_rows": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "srgpt_silu_mul": (i32, [vp, vp, i32, i32, i32, vp]),
    "srgpt_argmax": (i32, [vp, vp, i32, i32, vp]),
    "srgpt_cross_entropy": (i32, [vp, vp, vp, vp, i32, i32, i64, vp]),
    "srgpt_sample_ws_bytes": (i64, [i32]),
    "srgpt_sample": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "srgpt_sample_status": (i32, [vp, i32, vp]),
    "srgpt_image_resize_normalize": (i32, [vp, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, f32, i32, i32, vp]),
    "srgpt_mask_resize_nearest": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, vp, i32, vp]),
    "srgpt_mask_pad_resize": (i32, [vp, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, i32, vp]),
    "srgpt_vit_assemble_cls": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "srgpt_vit_ws_bytes": (i64, [C.POINTER(VitWeights), i32]),
    "srgpt_vit_forward": (i32, [C.POINTER(VitWeights), vp, vp, vp, i32, vp]),
    "srgpt_llm_ws_bytes": (i64, [C.POINTER(LlmWeights), i32, i32]),
    "srgpt_llm_prefill": (i32, [C.POINTER(LlmWeights), C.POINTER(LlmState), vp, i32, vp, vp, vp]),
    "srgpt_llm_prefill_ragged": (i32, [C.POINTER(LlmWeights), C.POINTER(LlmState), vp, i32, vp, vp, vp, vp]),
    "srgpt_llm_decode_step": (i32, [C.POINTER(LlmWeights), C.POINTER(LlmState), vp]),
    "srgpt_llm_sample_first": (i32, [C.POINTER(LlmWeights), C.POINTER(LlmState), vp]),
    "srgpt_llm_decode_sync_state": (i32, [C.POINTER(LlmWeights), C.POINTER(LlmState), vp]),
    "srgpt_llm_decode_graph_create": (i32, [C.POINTER(LlmWeights), C.POINTER(LlmState), vp, C.POINTER(vp)]),
    "srgpt_graph_launch": (i32, [vp, i32, vp]),
    "srgpt_graph_destroy": (i32, [vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: Optional[C.CDLL] = None


class SrgptNativeError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libsrgpt_hip.so (once).  Raises SrgptNativeError if it is missing -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SrgptNativeError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run "
            "`make -C spatialrgpt_amd/csrc` (needs hipcc; cross-compiles for gfx950 without a GPU). "
            "spatialrgpt_amd has no non-HIP fallback by design.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SrgptNativeError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SrgptNativeError(f"{LIB_PATH} does not export {name}; rebuild the extension") from e
        fn.restype = res
        fn.argtypes = args
    if lib.srgpt_abi_version() != ABI_VERSION:
        raise SrgptNativeError("libsrgpt_hip.so ABI version mismatch; rebuild the extension")
    _lib = lib
    return lib


def last_error() -> str:
    return load().srgpt_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map C return codes to the exception types the reference raises (SURVEY 8b error conventions)."""
    if rc == 0:
        return
    msg = last_error()
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(f"srgpt native error {rc}: {msg}")
