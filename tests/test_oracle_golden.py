"""Pin the CPU oracle against fixtures produced by the reference's own Python
(tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import hashlib

import numpy as np
import pytest

from oracle import quip_oracle as O


def test_e8p_packed_abs_grid_matches_reference(golden, golden_meta):
    mine = O.e8p_grid_packed_abs()
    assert mine.dtype == np.int64 and mine.shape == (256,)
    np.testing.assert_array_equal(mine, golden["e8p_grid_packed_abs"])
    assert hashlib.sha256(mine.tobytes()).hexdigest() == golden_meta["e8p_packed_sha256"]
    # anchors quoted in SURVEY.md appendix A.2
    u = mine.view(np.uint64)
    assert u[0] == 0x0202020202020202 and u[1] == 0xFA02020202020202
    assert u[226] == 0xFE0202020206020A and u[255] == 0xFE06060602060206


def test_e8p_decode_all_65536_codes(golden, golden_meta):
    full = O.e8p_full_grid_i8()
    assert full.shape == (65536, 8) and full.dtype == np.int8
    assert hashlib.sha256(full.tobytes()).hexdigest() == golden_meta["e8p_full_sha256"]
    np.testing.assert_array_equal(full[golden["e8p_full_sample_idx"]], golden["e8p_full_sample_i8"])
    # structural facts: all rows distinct, 4w odd in [-11, 11], spot values
    assert len(np.unique(full, axis=0)) == 65536
    assert set(np.unique(np.abs(full))) <= {1, 3, 5, 7, 9, 11, 13, 15}
    np.testing.assert_array_equal(full[0], [3] * 8)
    np.testing.assert_array_equal(full[1], [1] * 8)
    np.testing.assert_array_equal(full[0x0101], [1, 1, 1, 1, 1, 1, 1, -7])
    np.testing.assert_array_equal(full[0xFFFF], [-5, -5, -1, -1, -5, -5, -5, 3])


def test_e81b_tables(golden):
    g = O.e81b_grid()
    np.testing.assert_array_equal(g.astype(np.float32), golden["e81b_grid"])
    np.testing.assert_array_equal(O.e81b_grid_packed(g), golden["e81b_grid_packed"])
    # kernel-side unpack of the packed table gives back the grid (origin_order.cu:908-922)
    dec = O.e81b_decode_twice(np.arange(256)).astype(np.float64) / 2
    np.testing.assert_array_equal(dec, g)


def test_d4_grid(golden):
    np.testing.assert_array_equal(O.d4_grid().astype(np.float32), golden["d4_grid"])


def test_hi_pack_and_decode(golden):
    np.testing.assert_array_equal(O.hi_pack(golden["hi_idx"]), golden["hi_packed"])
    dense = O.decompress_hi(golden["hi_packed"]).astype(np.float32)
    np.testing.assert_array_equal(dense, golden["hi_dense"])


@pytest.mark.parametrize("n,K", [(256, 1), (688, 43), (4096, 1), (11008, 43), (1792, 7)])
def test_hadamard_matches_reference_butterfly(golden, n, K):
    x = golden[f"fht_{n}_x"]
    hadK = golden.get(f"fht_{n}_hadK")
    y = O.matmul_hadU(x, hadK, K, n)
    yt = O.matmul_hadU(x, hadK, K, n, transpose=True)
    np.testing.assert_allclose(y, golden[f"fht_{n}_y"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(yt, golden[f"fht_{n}_yt"], rtol=0, atol=2e-5)


def test_hadamard_is_orthogonal_and_sylvester():
    x = np.eye(8)
    h = O.fwht(x)
    ref = np.array([[(-1) ** bin(i & j).count("1") for j in range(8)] for i in range(8)], float)
    np.testing.assert_array_equal(h, ref)


def test_get_hadK_shapes(golden_meta):
    for key, (K, padn, shape) in golden_meta["get_hadK"].items():
        n, ur = key.split("_")
        n = int(n)
        if ur == "1":
            e, base = O.split_pow2(n)
            assert K == base and padn == n
            assert (shape is None) == (base == 1)


@pytest.mark.parametrize("cbid", ["E8P12", "E8P12RVQ4B", "E8P12RVQ3B"])
def test_quantize_goldens_pin_index_packing(golden, golden_meta, cbid):
    """The reference quantiser returns (values, indices); decoding its indices
    with the oracle must reproduce its values (fp32 there, one fp16 fma here)."""
    Q = golden[f"quant_{cbid}_Qidxs"]
    vals = golden[f"quant_{cbid}_vals"]
    scale = golden_meta.get(f"quant_{cbid}_resid_scale", 0.0)
    W = O.decompress(cbid, Q, scale).astype(np.float32)
    assert W.shape == vals.shape
    if cbid == "E8P12":
        np.testing.assert_array_equal(W, vals)
    else:
        # fp16(scale) and the fp16 rounding of the fma: |err| <= ulp16(4)/2 + |r|*2^-11*scale
        np.testing.assert_allclose(W, vals, rtol=0, atol=3e-3)


def _layer_from_case(case):
    return O.make_layer(case["codebook"], case["in_features"], case["out_features"],
                        seed=case["seed"], bias=case["bias"], per_channel=case["per_channel"],
                        resid_scale=case["resid_scale"])


def test_module_goldens_staged_and_exact(golden, golden_meta):
    """oracle QuantLinear.forward vs the reference's QuantLinear.forward."""
    for case in golden_meta["module_cases"]:
        P = _layer_from_case(case)
        What = O.qlinear_dense_weight(P)
        for M in case["Ms"]:
            x = golden[f"mod{case['idx']}_M{M}_x"]
            yref = golden[f"mod{case['idx']}_M{M}_y"].astype(np.float64)
            ystaged = O.qlinear_forward(P, x, "staged", What).astype(np.float64)
            yexact = O.qlinear_forward(P, x, "exact", What)
            tol = O.parity_bound(P, x, What)
            assert np.all(np.abs(yref - yexact) <= tol), (case, M)
            # staged pipeline rounds at the same points as the reference -> ~1 ulp agreement
            rms = np.sqrt((yexact ** 2).mean())
            assert np.max(np.abs(yref - ystaged)) <= 2.0 ** -9 * (np.abs(yexact).max() + rms), (case, M)


def test_calc_weight_goldens(golden, golden_meta):
    """Path identity: dense W = hadU_R(hadU_L(decode)^T) of the reference
    (qlinear.py:144-159) equals the oracle's operator."""
    for case in golden_meta["module_cases"]:
        key = f"mod{case['idx']}_W"
        if key not in golden:
            continue
        P = _layer_from_case(case)
        rows = golden[f"mod{case['idx']}_Wrows"]
        Wd = O.qlinear_dense_weight(P)                       # (q_out, q_in)
        A = O.matmul_hadU(Wd, P.had_left, P.K_left, P.q_in, P.wscale_float)   # rows: out
        Wfull = O.matmul_hadU(A.T, P.had_right, P.K_right, P.q_out)          # (q_in, q_out)
        ref = golden[key].astype(np.float64)
        scale = np.abs(Wfull).max()
        np.testing.assert_allclose(Wfull[rows], ref, rtol=0, atol=4 * 2.0 ** -11 * scale)


def test_config1_single_layer_4096(golden, golden_meta):
    """BASELINE.json configs[0]: E8P12 4096->4096 CPU decompress+matmul."""
    c = golden_meta["cfg1"]
    P = O.make_layer(c["codebook"], c["in_features"], c["out_features"], seed=c["seed"])
    x = golden["cfg1_x"]
    What = O.qlinear_dense_weight(P)
    y = O.qlinear_forward(P, x, "exact", What)
    tol = O.parity_bound(P, x, What)
    assert np.all(np.abs(golden["cfg1_y"].astype(np.float64) - y) <= tol)
    ys = O.qlinear_forward(P, x, "staged", What).astype(np.float64)
    assert np.max(np.abs(golden["cfg1_y"] - ys)) <= 2.0 ** -9 * np.abs(y).max()


def test_state_dict_layout(golden_meta):
    lay = golden_meta["state_dict_layout"]
    assert lay["E8P12"]["Qidxs"] == [[4096, 1376], "torch.int16"]
    assert lay["E8P12RVQ4B"]["Qidxs"] == [[4096, 1376], "torch.int32"]
    assert lay["E8P12RVQ3B"]["Qidxs"] == [[4096, 1032], "torch.int32"]
    assert lay["D4"]["Qidxs"] == [[4096, 2752], "torch.uint8"]
    assert lay["HI"]["Qidxs"] == [[4096, 1376], "torch.int32"]
    for cb in lay:
        assert O.qidx_cols(cb, 11008) == lay[cb]["Qidxs"][0][1]
        assert lay[cb]["_K"] == [43, 1, 11008, 4096]
        assert lay[cb]["had_left"] == [[43, 43], "torch.float16"]
