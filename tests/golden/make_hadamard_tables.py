#!/usr/bin/env python3
"""Bundle the +-1 Hadamard factors the reference ships as a data file (hadamard.safetensors, opened by
quant.py:8 and used by get_hadK(use_rand=False), quant.py:34-39) as bit-packed arrays:

    quip_for_all_amd/data/hadamard_tables.npz   orders (int32[]), bits_<order> (uint8, np.packbits of H > 0, row major)
    tests/golden/hadamard_tables.json           sha256 of every matrix as int8 (+1 / -1), row major

Run in the authoring container (needs /root/reference):  python tests/golden/make_hadamard_tables.py
The tables are data (Hadamard matrices of order 4 * odd <= 252 plus 1, 2, 4), not source text.  Every matrix is checked
to satisfy H H^T = n I before it is written; tests/test_hadamard_tables.py re-checks that and the hashes on
every run."""
import hashlib
import json
import os

import numpy as np

REF = "/root/reference/hadamard.safetensors"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def main():
    from safetensors import safe_open
    f = safe_open(REF, "np")
    out, meta = {}, {}
    orders = sorted(int(k) for k in f.keys())
    for n in orders:
        H = f.get_tensor(str(n))
        assert H.shape == (n, n) and set(np.unique(H).tolist()) <= {-1.0, 1.0}, n
        Hi = H.astype(np.int64)
        assert np.array_equal(Hi @ Hi.T, n * np.eye(n, dtype=np.int64)), n
        out[f"bits_{n}"] = np.packbits((H > 0).reshape(-1))
        meta[str(n)] = hashlib.sha256(H.astype(np.int8).tobytes()).hexdigest()
    out["orders"] = np.asarray(orders, dtype=np.int32)
    dst = os.path.join(REPO, "quip_for_all_amd", "data", "hadamard_tables.npz")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    np.savez_compressed(dst, **out)
    with open(os.path.join(HERE, "hadamard_tables.json"), "w") as g:
        json.dump({"provenance": "reference data file hadamard.safetensors (quant.py:8)", "sha256_int8": meta}, g,
                  indent=1, sort_keys=True)
    print("orders", orders, "->", os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
