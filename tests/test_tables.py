"""Product-side constant tables vs the reference-generated goldens (CPU only)."""
import hashlib

import numpy as np

from quip_for_all_amd.codebook import tables


def test_e8p_packed_abs(golden, golden_meta):
    t = tables.e8p_grid_packed_abs()
    np.testing.assert_array_equal(t, golden["e8p_grid_packed_abs"])
    assert hashlib.sha256(t.tobytes()).hexdigest() == golden_meta["e8p_packed_sha256"]


def test_e8p_full_grid(golden, golden_meta):
    g = tables.e8p_full_grid()
    i8 = np.round(g * 4).astype(np.int8)
    assert hashlib.sha256(i8.tobytes()).hexdigest() == golden_meta["e8p_full_sha256"]


def test_e81b(golden):
    np.testing.assert_array_equal(tables.e81b_grid(), golden["e81b_grid"])
    np.testing.assert_array_equal(tables.e81b_grid_packed(), golden["e81b_grid_packed"])


def test_d4(golden):
    np.testing.assert_array_equal(tables.d4_grid(), golden["d4_grid"])
    assert hashlib.sha256(tables.d4_grid().astype(np.float16).tobytes()).hexdigest() == \
        "3055b7ccb5181fb734c0f0a5bf566f79481d6efbaf0bcc24c9dbe259c95b7968"


def test_hi_pack_matches_reference(golden):
    import torch
    from quip_for_all_amd.codebook import HI4B1C_codebook
    cb = HI4B1C_codebook(inference=True)
    packed = cb.maybe_pack_idxs(torch.from_numpy(golden["hi_idx"]))
    np.testing.assert_array_equal(packed.numpy(), golden["hi_packed"])


def test_rvq3_pack_matches_oracle():
    import torch
    from oracle import quip_oracle as O
    from quip_for_all_amd.codebook import E8P12RVQ3B_codebook
    cb = E8P12RVQ3B_codebook(inference=True)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 1 << 24, (4, 32)).astype(np.int32)
    np.testing.assert_array_equal(cb.maybe_pack_idxs(torch.from_numpy(idx)).numpy(), O.rvq3_pack(idx))
