"""The C restatement of the oracle agrees with the numpy oracle (which is pinned to the
reference-generated goldens)."""
import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import quip_oracle as O


def test_c_decode_matches_all_codes():
    q = np.arange(65536, dtype=np.uint32).astype(np.uint16).reshape(256, 256)
    w = C.decompress_e8p(q, O.e8p_grid_packed_abs())
    ref = O.e8p_full_grid_i8().astype(np.float32) / 4
    np.testing.assert_array_equal(w.reshape(65536, 8), ref)


def test_c_fwht():
    x = np.random.default_rng(0).standard_normal(1024).astype(np.float32)
    np.testing.assert_allclose(C.fwht(x), O.fwht(x.astype(np.float64)), rtol=0, atol=1e-3)


@pytest.mark.parametrize("fin,fout,bias", [(256, 256, False), (688, 256, True), (256, 688, False), (1792, 512, False)])
def test_c_qlinear_forward(fin, fout, bias):
    P = O.make_layer("E8P12", fin, fout, seed=fin + fout, bias=bias)
    x = np.random.default_rng(1).standard_normal((1, fin)).astype(np.float16)
    y = C.qlinear_forward(P, x.astype(np.float32))
    ref = O.qlinear_forward(P, x, "exact")[0]
    assert np.max(np.abs(y - ref)) <= 1e-4 * (np.abs(ref).max() + 1)


def test_c_oracle_against_reference_module_golden(golden, golden_meta):
    case = golden_meta["module_cases"][0]
    P = O.make_layer(case["codebook"], case["in_features"], case["out_features"], seed=case["seed"],
                     bias=case["bias"], per_channel=case["per_channel"], resid_scale=case["resid_scale"])
    x = golden["mod0_M1_x"]
    y = C.qlinear_forward(P, x.astype(np.float32))
    tol = O.parity_bound(P, x)[0]
    assert np.all(np.abs(y - golden["mod0_M1_y"].astype(np.float64)[0]) <= tol)
