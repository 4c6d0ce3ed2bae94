"""The C-ABI library builds (cross-compiled for gfx950), loads without a GPU and
exports every symbol include/quip_mi355.h declares.  No compute calls here."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "quip_mi355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(quip_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def libpath():
    import __graft_entry__ as g
    return g.build_library(verbose=False)


def test_header_declares_reference_boundary():
    names = _declared()
    for n in ("quip_hadamard_f16", "quip_e8p_mm_origorder", "quip_e8prvq3_mm_origorder",
              "quip_e8prvq4_mm_origorder", "quip_d4_mm_origorder", "quip_hi_mm_origorder",
              "quip_decompress_e8p_origorder", "quip_decompress_e8prvq3_origorder",
              "quip_decompress_e8prvq4_origorder", "quip_decompress_d4_origorder",
              "quip_decompress_hi_origorder"):
        assert n in names


def test_library_exports_every_declared_symbol(libpath):
    L = ctypes.CDLL(libpath)
    for n in _declared():
        assert hasattr(L, n), f"{n} declared in include/quip_mi355.h but not exported"
    L.quip_abi_version.restype = ctypes.c_int
    hdr = open(os.path.join(REPO, 'include', 'quip_mi355.h')).read()
    assert L.quip_abi_version() == int(re.search(r'#define QUIP_ABI_VERSION (\d+)', hdr).group(1))
    L.quip_strerror.restype = ctypes.c_char_p
    assert L.quip_strerror(0) == b"ok" and b"null" in L.quip_strerror(-1)


def test_python_binding_covers_header(libpath):
    from quip_for_all_amd import capi
    assert set(_declared()) - {"quip_strerror"} <= set(capi.SIGNATURES)
    capi.check_symbols()


def test_argument_validation_without_gpu(libpath):
    """error paths return codes before any launch, so they are testable on CPU"""
    from quip_for_all_amd import capi
    L = capi.lib()
    assert L.quip_e8p_mm_origorder(None, None, None, None, 1, 8, 8, None) == -1      # null
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    assert L.quip_e8p_mm_origorder(p16, p16, p16, p16, 1, 8, 12, None) == -2          # k % 8
    assert L.quip_e8prvq3_mm_origorder(p16, p16, p16, p16, 0.5, p16, 1, 8, 8, None) == -2  # k % 32
    assert L.quip_e8p_mm_origorder(p16 + 2, p16, p16, p16, 2, 8, 8, None) == -3       # misaligned x
    assert L.quip_hadamard_f16(p16, p16, 1, 24, 1.0, None) == -2                      # not a power of two
    assert L.quip_decompress_hi_origorder(p16, p16, 0, 8, None) == 0                  # empty: ok, no launch
    assert L.quip_hi_mm_origorder(p16, p16, p16, 0, 8, 8, None) == 0                  # m == 0: ok
    # round 3 entry points: the fused tile kernel's codebook modes, the attention launch's window, the D4 table GEMV
    assert L.quip_e8prvq4_mm_batched(None, p16, p16, 0.3, p16, 40, 64, 64, None) == -1
    assert L.quip_e8prvq4_mm_batched(p16, p16, None, 0.3, p16, 40, 64, 64, None) == -1
    assert L.quip_e8prvq3_mm_batched(p16, p16, p16, None, 0.3, p16, 40, 64, 64, None) == -1
    assert L.quip_d4_mm_batched(p16, p16, p16, p16 + 2, 40, 64, 64, None) == -3          # misaligned y
    assert L.quip_hi_mm_batched(p16, p16, p16, -1, 64, 64, None) == -2                   # negative m
    assert L.quip_hi_mm_batched(p16, p16, p16, 0, 64, 64, None) == 0                     # empty: ok, no launch
    assert L.quip_rope_attn_decode_window_f16(None, p16, p16, p16, p16, p16, p16, p16, p16, 4, 4, 64, 32, 0.125, 8, None, None) == -1
    assert L.quip_rope_attn_decode_window_f16(p16, p16, p16, p16, p16, p16, p16, p16, p16, 4, 4, 64, 32, 0.125, -1, None, None) == -2
    vp = (ctypes.c_void_p * 1)(p16)
    n1 = (ctypes.c_int32 * 1)(8)
    assert L.quip_d4_gemv_planes_group_ws(vp, vp, None, vp, n1, 1, 128, None, 0, None) == -1
    assert L.quip_d4_gemv_planes_group_ws(vp, vp, p16, vp, n1, 1, 12, None, 0, None) == -2      # k % 8
    assert L.quip_d4_gemv_planes_v2(p16, p16, p16, None, 8, 128, None, 0, None) == -1
    # round 5: the shape-1 launch's re-tiled copy of a code matrix
    assert L.quip_tile_codes(None, p16, 16, 64, None) == -1
    assert L.quip_tile_codes(p16, p16 + 1024, 24, 64, None) == -2                        # rows % 16
    assert L.quip_tile_codes(p16, p16 + 1024, 16, 96, None) == -2                        # row_bytes % 64
    assert L.quip_tile_codes(p16, p16 + 1026, 16, 64, None) == -3                        # misaligned destination
    assert L.quip_tile_codes(p16, p16, 16, 64, None) == -5                               # in place
    assert L.quip_tile_codes(p16, p16 + 512, 16, 64, None) == -5                         # partial overlap (16 x 64 = 1024 bytes each)
    assert L.quip_tile_codes(p16 + 512, p16, 16, 64, None) == -5
    assert L.quip_tile_codes(p16, p16 + 1024, 0, 64, None) == 0                          # empty: ok, no launch
    assert L.quip_untile_codes(p16, None, 16, 64, None) == -1
    assert L.quip_untile_codes(p16, p16 + 1024, 24, 64, None) == -2
    assert L.quip_untile_codes(p16, p16 + 512, 16, 64, None) == -5                       # overlap
    assert L.quip_untile_codes(p16, p16 + 1024, 0, 64, None) == 0


def test_ops_registered_and_fail_loudly_on_cpu(libpath):
    import torch
    import quip_for_all_amd  # noqa: F401
    for name in ("hadamard", "e8p_mm_origorder", "e8prvq3_mm_origorder", "e8prvq4_mm_origorder",
                 "d4_mm_origorder", "hi_mm_origorder", "decompress_e8p_origorder",
                 "decompress_e8prvq3_origorder", "decompress_e8prvq4_origorder",
                 "decompress_d4_origorder", "decompress_hi_origorder"):
        assert hasattr(torch.ops.quip_lib, name)
    x = torch.zeros(1, 8, dtype=torch.float16)
    q = torch.zeros(4, 1, dtype=torch.int16)
    g = torch.zeros(256, dtype=torch.int64)
    with pytest.raises((NotImplementedError, RuntimeError)):   # no CPU kernel, no fallback
        torch.ops.quip_lib.e8p_mm_origorder(x, q, g)
    # fake tensors: shape inference only
    y = torch.ops.quip_lib.e8p_mm_origorder(x.to("meta"), q.to("meta"), g.to("meta"))
    assert y.shape == (1, 4) and y.dtype == torch.float16


def test_quantlinear_layout_matches_reference(golden_meta):
    import torch
    from quip_for_all_amd.codebook import codebook_id
    from quip_for_all_amd.qlinear import QuantLinear
    for cbid, lay in golden_meta["state_dict_layout"].items():
        L = QuantLinear(11008, 4096, codebook_id[cbid](inference=True), bias=True, use_rand=True)
        mine = {k: [list(v.shape), str(v.dtype)] for k, v in L.state_dict().items()}
        ref = {k: v for k, v in lay.items() if not k.startswith("_")}
        assert mine == ref, cbid
        assert [L.K_left, L.K_right, L.q_in_features, L.q_out_features] == lay["_K"]


def test_product_fails_loudly_without_the_native_library(tmp_path):
    """no CPU fallback: with the library missing the first native call raises QuipNativeError (fresh interpreter,
    QUIP_LIB_PATH pointing at nothing)"""
    import subprocess
    import sys
    code = ("import quip_for_all_amd.capi as c\n"
            "try:\n    c.lib()\nexcept c.QuipNativeError as e:\n    print('RAISED', 'no CPU fallback' in str(e).lower() or 'missing' in str(e))\n"
            "else:\n    print('LOADED')\n")
    env = dict(os.environ, QUIP_LIB_PATH=str(tmp_path / "nope.so"), PYTHONPATH=REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=REPO, timeout=300)
    assert "RAISED True" in out.stdout, (out.stdout, out.stderr[-500:])


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under quip_for_all_amd/ may import it"""
    import re
    pkg = os.path.join(REPO, "quip_for_all_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(root, f)
