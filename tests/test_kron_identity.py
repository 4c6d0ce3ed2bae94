"""The identity the Llama-3-8B / Mistral-7B instantiation of the 4096-wide persistent launch rests on (csrc/decode_block.hip,
QUIP_BLOCK_G8): the reference's randomised Hadamard transform of a 14336-vector on the (7, 2048) view (quant.py:26-39, 72-88:
U = (R_7 (x) H_2048) / sqrt 2048) IS the (56, 256)-view transform ((R_7 (x) H_8) (x) H_256) / (sqrt 8 sqrt 256) -- so a launch
built for K x 256 views with a dense K x K factor serves it with the factor R_7 (x) H_8.  Checked on the CPU oracle's own
matmul_hadU (pinned on the reference's goldens, tests/test_oracle_golden.py)."""
import numpy as np

from oracle import quip_oracle as O


def _sylvester(n):
    h = np.array([[1.0]])
    while h.shape[0] < n:
        h = np.block([[h, h], [h, -h]])
    return h


def test_7x2048_transform_equals_56x256_transform_with_kron_factor():
    rng = np.random.default_rng(5)
    r7, _ = np.linalg.qr(rng.standard_normal((7, 7)))
    x = rng.standard_normal((3, 14336))
    # the reference's form: K = 7, the 2048-point transform over the last axis, then R_7 over the first
    ref = O.matmul_hadU(x, r7, 7, 14336)
    # the launch's form: K = 56 with the factor R_7 (x) H_8 / sqrt 8, 256-point transforms
    m56 = np.kron(r7, _sylvester(8)) / np.sqrt(8.0)
    got = O.matmul_hadU(x, m56, 56, 14336)
    assert np.allclose(got, ref, rtol=0, atol=1e-10)
    # and the transposed (input side) form, quant.py:79-80
    ref_t = O.matmul_hadU(x, r7.T, 7, 14336)
    got_t = O.matmul_hadU(x, m56.T, 56, 14336)
    assert np.allclose(got_t, ref_t, rtol=0, atol=1e-10)


def test_kron_factor_entries_are_exact_in_fp16_when_r7_is():
    """what the host hands the launch: R_7 (x) H_8 WITHOUT the 1 / sqrt 8 (the kernel folds it into its scales): entries +-R_7"""
    r7 = np.random.default_rng(6).standard_normal((7, 7)).astype(np.float16)
    m = np.kron(r7.astype(np.float64), _sylvester(8))
    assert np.array_equal(m.astype(np.float16).astype(np.float64), m)
