"""Host-side pin of the kernel regimes (which product path a QuantLinear forward takes, by codebook, row count and shape)
for every BASELINE shape under the DEFAULT environment: a threshold edit that silently moves one of these shapes to a
slower kernel fails here.  No GPU: the decisions are host code (QuantLinear.regime, codebook.batched_regime,
quip_e8p_gemv_kernel_choice)."""
import ctypes
import os

import pytest
import torch

SHAPES_7B = [(4096, 4096), (4096, 11008), (11008, 4096)]          # (in, out)
SHAPES_70B = [(8192, 8192), (8192, 1024), (8192, 28672), (28672, 8192)]


@pytest.fixture(autouse=True)
def _default_env():
    # the class-level switches are read from the environment when the package is imported: the table below is the one of the
    # DEFAULT environment (reloading the modules here instead would leave other tests with stale class objects)
    odd = [k for k in os.environ if k.startswith("QUIP_") and k not in ("QUIP_LIB_PATH", "QUIP_HADAMARD_TABLES")]
    if odd:
        pytest.skip("non-default environment: %s" % ", ".join(odd))


def _layer(cb_name, fin, fout):
    import quip_for_all_amd.qlinear as Q
    from quip_for_all_amd.codebook import codebook_id
    cb = codebook_id[cb_name](inference=True)
    with torch.device("meta"):
        layer = Q.QuantLinear(fin, fout, cb, bias=False, use_rand=True)
    return layer


@pytest.mark.parametrize("fin,fout", SHAPES_7B + SHAPES_70B)
def test_e8p12_regimes_by_row_count(fin, fout):
    layer = _layer("E8P12", fin, fout)
    cb = layer.codebook
    rpp = layer._rows_per_pass()
    assert rpp == (5 if fin <= 4096 else 3 if fin <= 8192 else 2 if fin <= 15360 else 1)
    assert layer.regime(1) == "gemv_planes"
    for m in range(2, 32):
        want = "rows_exact" if m <= rpp else "skinny_fp16"
        assert layer.regime(m) == want, (m, layer.regime(m))
    for m in (32, 64, 256, 2048, 32768):
        assert layer.regime(m) == "codebook"
        # beyond the skinny regime the default is the reference's shape, decompress + vendor GEMM: faster than the fused
        # kernel at every M on this part (profiles/r03_prefill_crossover.txt); QUIP_BATCHED_MM=fused selects the latter
        want = "skinny_chunks" if m * fout <= 1_200_000 else "decompress_gemm"
        assert cb.batched_regime(m, fout, fin) == want, (m, cb.batched_regime(m, fout, fin))


def test_bs1_gemv_kernel_by_launch():
    """the launches of a decode step: which of the two matrix-core GEMV kernels each one takes"""
    from quip_for_all_amd import capi
    L = capi.lib()

    def choice(ns, k):
        arr = (ctypes.c_int32 * len(ns))(*ns)
        return L.quip_e8p_gemv_kernel_choice(arr, len(ns), k)
    # Llama-2-7B: q/k/v group, o, gate/up group (22.5 MB: the K-splitting kernel), down
    assert choice([4096, 4096, 4096], 4096) == 1
    assert choice([4096], 4096) == 1
    assert choice([11008, 11008], 4096) == 2
    assert choice([4096], 11008) == 1
    # Llama-2-70B: q/k/v group (21 MB at k = 8192), o (16.8 MB at k = 8192), gate/up group, down (k = 28672)
    assert choice([8192, 1024, 1024], 8192) == 2
    assert choice([8192], 8192) == 2
    assert choice([28672, 28672], 8192) == 2
    assert choice([28672], 8192) == 2
    assert choice([8192], 28672) == 2
    assert choice([1024], 8192) == 1


@pytest.mark.parametrize("cb_name", ["E8P12RVQ4B", "E8P12RVQ3B", "D4", "HI"])
def test_other_codebooks_regimes(cb_name):
    """bs = 1 on the matrix-core GEMV's table modes, 2..5 rows in exact rows mode, up to a few hundred rows on the single-pass
    skinny kernel, larger batches through the codebook's decompress + dense GEMM (as the reference does)"""
    layer = _layer(cb_name, 4096, 4096)
    assert layer.regime(1) == "gemv_planes"
    for m in (2, 5, 16, 31):
        # E8P12RVQ4B beyond one exact pass: the single-pass fp16 skinny kernel in its RVQ4 mode (round 3)
        # beyond one exact pass: the single-pass fp16 skinny kernel in the codebook's mode (round 3)
        want = "skinny_fp16" if m > 5 else "rows_exact"
        assert layer.regime(m) == want, (m, layer.regime(m))
    for m in (32, 2048):
        assert layer.regime(m) == "codebook"
    assert layer.codebook.batched_regime(32, 4096, 4096) == "skinny_chunks"
    assert layer.codebook.batched_regime(2048, 4096, 4096) == "decompress_gemm"
    # QUIP_BATCHED_MM=fused: the fused dequant + MFMA tile kernel in the codebook's mode beyond the skinny regime
    from quip_for_all_amd.codebook.codebooks import E8P12_codebook
    saved = E8P12_codebook.batched_mode
    try:
        E8P12_codebook.batched_mode = "fused"
        assert layer.codebook.batched_regime(2048, 4096, 4096) == "fused_gemm"
        assert layer.codebook.batched_regime(32, 4096, 4096) == "skinny_chunks"
    finally:
        E8P12_codebook.batched_mode = saved
    # Llama-2-70B down_proj width: every codebook's bs = 1 product stays on the matrix-core GEMV (the RVQ codebooks' and
    # HI's 2k = 57344-wide virtual rows through the K-splitting kernel)
    assert layer.codebook.planes_supported(8192, 28672)


def test_persistent_launch_shapes():
    """the decode engine's launches take exactly the Llama-2-7B block (stage 2) / the MLP shapes with an instantiation"""
    from quip_for_all_amd import capi
    L = capi.lib()
    assert L.quip_ffn_engine_workspace_bytes(11008, 43) > 0
    assert L.quip_block_engine_layer_bytes() == 256
