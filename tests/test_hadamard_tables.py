"""use_rand=False Hadamard factors (reference quant.py:8,34-39): the bundled bit-packed tables
are exactly the reference's data (sha256 pins written by tests/golden/make_hadamard_tables.py), every one is a
Hadamard matrix, and get_hadK reproduces the reference's (K, padded n) for the Llama sizes (reference_golden.json
`get_hadK`, produced by the imported reference)."""
import hashlib
import json
import math
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _tables():
    from quip_for_all_amd import quant
    return quant._had_tables()


def test_tables_are_the_references_and_hadamard():
    with open(os.path.join(HERE, "golden", "hadamard_tables.json")) as f:
        pins = json.load(f)["sha256_int8"]
    tabs = _tables()
    assert sorted(tabs, key=int) == sorted(pins, key=int)
    for k, H in tabs.items():
        n = int(k)
        Hn = H.numpy()
        assert Hn.shape == (n, n)
        assert hashlib.sha256(Hn.astype(np.int8).tobytes()).hexdigest() == pins[k], k
        Hi = Hn.astype(np.int64)
        assert np.array_equal(Hi @ Hi.T, n * np.eye(n, dtype=np.int64)), k


def test_get_hadK_matches_reference_shapes():
    from quip_for_all_amd.quant import get_hadK
    with open(os.path.join(HERE, "golden", "reference_golden.json")) as f:
        ref = json.load(f)["get_hadK"]
    for key, (K, padn, shape) in ref.items():
        n, ur = (int(v) for v in key.split("_"))
        m, k2, p2 = get_hadK(n, bool(ur))
        assert (k2, p2) == (K, padn), key
        assert (None if m is None else list(m.shape)) == shape, key
        if m is not None:   # orthonormal factor in both modes
            assert torch.allclose(m @ m.T, torch.eye(K), atol=1e-5), key
    m, K, n = get_hadK(11008, use_rand=False)
    assert K == 172 and n == 11008 and torch.equal(m * math.sqrt(172), _tables()["172"])
    assert get_hadK(28672, use_rand=False)[1] == 28
    # no table for this base (4 * 257 > 252) and exp < 2: padded to the next power of two like the reference
    assert get_hadK(2 * 4 * 257, use_rand=False)[1:] == (1, 4096)
    assert get_hadK(2 * 43, use_rand=False)[1:] == (1, 128)


def test_missing_table_file_raises(monkeypatch, tmp_path):
    from quip_for_all_amd import quant
    monkeypatch.setattr(quant, "_HAD_TABLES", None)
    monkeypatch.setattr(quant, "_HAD_TABLES_FILE", str(tmp_path / "nope.npz"))
    with pytest.raises(FileNotFoundError):
        quant.get_hadK(11008, use_rand=False)
    monkeypatch.setenv("QUIP_HADAMARD_TABLES", str(tmp_path / "also_nope.safetensors"))
    with pytest.raises(FileNotFoundError):
        quant.get_hadK(11008, use_rand=False)
    monkeypatch.setattr(quant, "_HAD_TABLES", None)   # leave the module clean for later tests
