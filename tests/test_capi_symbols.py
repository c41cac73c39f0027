"""CPU: the C-ABI library builds/loads here (hipcc cross-compiles) and exports every symbol include/srgpt.h
declares.  No compute call is made (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from tests.util import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "srgpt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(srgpt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from spatialrgpt_amd import _lib

    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/srgpt.h but not exported"
    assert set(names) == set(_lib.EXPORTED_SYMBOLS), set(names) ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.srgpt_abi_version() == _lib.ABI_VERSION == 9


def test_product_library_exports_nothing_but_the_header():
    """The product library carries no experiment residue: every exported `srgpt_*` C symbol is declared in include/srgpt.h (cross-file
    helpers have hidden visibility), and nothing of the tuning build (debug stamps, environment knobs, the VALU pooling kernel's
    bf16 instances) is in it."""
    import subprocess

    from spatialrgpt_amd import _lib

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("srgpt_")}
    assert exported == set(_declared()), exported ^ set(_declared())
    blob = open(_lib.LIB_PATH, "rb").read()
    for residue in (b"debug_stamps", b"SRGPT_GEMM_", b"SRGPT_REGION_", b"SRGPT_DECODE_", b"SRGPT_SKINNY_", b"SRGPT_GEMV_",
                    b"region_pool_kernelIDF16b"):
        assert residue not in blob, residue


def test_struct_layouts_match_header():
    """field order of the ctypes mirrors == field order in the header (guards silent ABI drift)."""
    from spatialrgpt_amd import _lib

    src = open(os.path.join(ROOT, "include", "srgpt.h")).read()

    def fields(struct_name):
        body = re.search(r"typedef struct \{([^}]*)\} " + struct_name + ";", src).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                out.append(re.findall(r"([A-Za-z_0-9]+)\s*$", part.strip())[0])
        return out

    assert fields("srgpt_vit_weights") == [f[0] for f in _lib.VitWeights._fields_]
    assert fields("srgpt_llm_weights") == [f[0] for f in _lib.LlmWeights._fields_]
    assert fields("srgpt_llm_state") == [f[0] for f in _lib.LlmState._fields_]
    assert fields("srgpt_sampling") == [f[0] for f in _lib.Sampling._fields_]
    assert ctypes.sizeof(_lib.Sampling) == 40  # the device block the host copies as raw bytes


def test_error_code_mapping():
    from spatialrgpt_amd import _lib

    _lib.check(0)
    for code, exc in ((_lib.ERR_ARG, ValueError), (_lib.ERR_UNSUPPORTED, NotImplementedError), (_lib.ERR_LAUNCH, RuntimeError)):
        with pytest.raises(exc):
            _lib.check(code)


def test_argument_validation_without_gpu():
    """argument errors are detected on the host before any launch -> safe to exercise without a GPU."""
    from spatialrgpt_amd import _lib

    lib = _lib.load()
    rc = lib.srgpt_gemm(None, None, None, None, None, 1, 1, 8, 8, 1, 0, 0, 0, 0, 0, 0, None, 0, _lib.BF16, None)
    assert rc == _lib.ERR_ARG and b"null" in lib.srgpt_last_error()
    rc = lib.srgpt_gemm(16, 16, None, None, 16, 4, 4, 7, 7, 4, 0, 0, 0, 0, 0, 0, None, 0, _lib.BF16, None)
    assert rc == _lib.ERR_ARG and b"multiples" in lib.srgpt_last_error()
    rc = lib.srgpt_gemv(16, 16, None, 0.0, None, 16, 1, 8, 12, 0, 0, _lib.BF16, None)
    assert rc == _lib.ERR_ARG and b"multiple of 8" in lib.srgpt_last_error()
    rc = lib.srgpt_gemv(16, 16, None, 0.0, 16, 16, 1, 8, 8, 1, 0, _lib.BF16, None)  # swiglu + residual
    assert rc == _lib.ERR_ARG and b"swiglu" in lib.srgpt_last_error()
    rc = lib.srgpt_gemv_w8(16, 16, None, None, 0.0, None, 16, 1, 8, 8, 0, 0, None)  # missing row scales
    assert rc == _lib.ERR_ARG and b"null" in lib.srgpt_last_error()
    # ABI 8: the row-statistics products validate before they ask the device anything
    rc = lib.srgpt_gemv_rowss(16, None, None, None, None, 0.0, None, 16, 4, 8, 8, 0, 0, None, None, 0, None)  # neither bf16 nor fp8 weights
    assert rc == _lib.ERR_ARG and b"null" in lib.srgpt_last_error()
    rc = lib.srgpt_gemv_rowss(16, None, 16, None, None, 0.0, None, 16, 4, 8, 8, 0, 0, None, None, 0, None)  # fp8 bytes without row scales
    assert rc == _lib.ERR_ARG and b"row scales" in lib.srgpt_last_error()
    # ABI 9: the packed decode layout
    rc = lib.srgpt_gemv_rowss(16, None, 16, 16, None, 0.0, None, 16, 4, 8, 8, 0, 0, None, None, 5, None)  # 5 rows per granule
    assert rc == _lib.ERR_ARG and b"packed_rows" in lib.srgpt_last_error()
    rc = lib.srgpt_gemv_rowss(16, None, 16, 16, None, 0.0, None, 16, 4, 16, 96, 0, 0, None, None, 4, None)  # fp8: K % 64
    assert rc == _lib.ERR_ARG and b"packed layout" in lib.srgpt_last_error()
    assert lib.srgpt_packed_bytes(50, 128, 1, 16) == 64 * 128 and lib.srgpt_packed_bytes(50, 128, 2, 4) == 52 * 128 * 2
    rc = lib.srgpt_pack_decode_weights(16, 16, 16, 48, 1, 4, None)  # fp8: K % 64
    assert rc == _lib.ERR_ARG and b"multiple of 64" in lib.srgpt_last_error()
    assert lib.srgpt_decode_attn_ws_floats(1, 32, 128) == 32 * 64 * 130 + 32  # split partials + arrival tickets
    rc = lib.srgpt_gemm_w8(16, 16, None, None, None, 16, 4, 4, 64, 64, 4, 0, 0, None, 0, None)  # missing row scales
    assert rc == _lib.ERR_ARG and b"null" in lib.srgpt_last_error()
    rc = lib.srgpt_gemm_w8(16, 16, 16, None, None, 16, 4, 4, 72, 64, 4, 0, 0, None, 0, None)  # lda < K
    assert rc == _lib.ERR_ARG and b"lda" in lib.srgpt_last_error()
    # v [M, L] + psum [M, ceil(L / 1024)] + partials [ceil(L / 300) row slabs, M, C] + one ticket per channel slab (fp32 worst case)
    assert lib.srgpt_region_pool_ws_floats(8, 108, 1152) == 8 * 11664 + 8 * 12 + 39 * 8 * 1152 + 36
    assert lib.srgpt_gemm_ws_bytes(259, 4096) == 8 * 259 * 4096 * 4
    # sampling: 128 slices x 64 candidates x (key + index) per sequence, the Gumbel-mode slice maxima, one error word
    assert lib.srgpt_sample_ws_bytes(2) == 2 * 128 * 64 * 8 + 2 * 128 * 8 + 256
    rc = lib.srgpt_sample(16, 16, 16, 16, 1, 128 * 2048 + 1, None)  # vocabulary beyond 128 slices x 2048 entries
    assert rc == _lib.ERR_UNSUPPORTED and b"vocabulary" in lib.srgpt_last_error()
    # ABI 7: the fused prefill entry points validate on the host before any launch
    rc = lib.srgpt_gemm_norm(16, 16, None, None, 16, 4, 8, 8, None, 0, 3, 16, None, 32, 1e-5, _lib.BF16, None)  # unknown norm kind
    assert rc == _lib.ERR_ARG and b"norm kind" in lib.srgpt_last_error()
    rc = lib.srgpt_gemm_norm(16, 16, None, None, 16, 4, 8, 8, None, 0, _lib.NORM_LAYER, 16, None, 32, 1e-5, _lib.BF16, None)  # LayerNorm without its bias
    assert rc == _lib.ERR_ARG and b"null" in lib.srgpt_last_error()
    rc = lib.srgpt_gemm_norm(16, 16, None, None, 32, 4, 8, 8, None, 0, _lib.NORM_RMS, 16, None, 32, 1e-5, _lib.BF16, None)  # Y aliases C
    assert rc == _lib.ERR_ARG and b"alias" in lib.srgpt_last_error()
    rc = lib.srgpt_gemm_rope_kv_append(16, 16, 16, 64, None, 0, None, 16, None, 16, 16, 1, 4, 2, 1, 32, 8, _lib.BF16, None)  # no k cache
    assert rc == _lib.ERR_ARG and b"null" in lib.srgpt_last_error()
    rc = lib.srgpt_gemm_rope_kv_append(16, 16, 16, 64, None, 0, 16, 16, None, 16, 16, 1, 9, 2, 1, 32, 8, _lib.BF16, None)  # T > max_pos
    assert rc == _lib.ERR_ARG and b"bad shape" in lib.srgpt_last_error()
    rc = lib.srgpt_gemm_swiglu(16, None, 16, 4, 64, 64, None, None, 0, _lib.BF16, None)
    assert rc == _lib.ERR_ARG and b"null" in lib.srgpt_last_error()
    rc = lib.srgpt_gemm_swiglu(16, 16, 16, 4, 64, 64, None, None, 0, _lib.BF16, None)  # a shape that needs the [M, 2 I] scratch
    assert rc == _lib.ERR_ARG and b"scratch" in lib.srgpt_last_error()


def test_public_header_is_self_contained_c_and_cxx(tmp_path):
    """include/srgpt.h is what a maintainer's cgo / JNI / ctypes-gen / C++ binding includes: it must compile on its own as C99 and as
    C++17 (round 6: `size_t` had come in without <stddef.h>)."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cc, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++17", "cpp")):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not installed")
        src = tmp_path / f"hdr.{ext}"
        src.write_text('#include "include/srgpt.h"\nint main(void) { return srgpt_abi_version == 0; }\n')
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-I", root, str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
