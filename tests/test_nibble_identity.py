"""Round 6: the nibble mode of the E8P12 decode (csrc/e8p_gemv_core.hip.h, "Nibble mode") restated in numpy and held against the
oracle on ALL 65 536 codes: the two 4-byte tables, the XOR identity, and the integer identity the matrix-core path rests on

    8 * sum_k digit[k] * 4w[k]  ==  (R1 - R2)[hi rows] + 16 * R2[lo rows] + 8 * SX[hi] - 120 * SX[lo]

with R1 = A x raw bytes, R2 = A x (raw & 0x0f) (int8 x int8 -> int32, exactly what v_mfma_i32_16x16x64_i8 computes).  CPU only:
the GPU tests hold the kernels themselves against the oracle (tests/test_gpu_exhaustive_codes.py, test_gpu_block_engine_gqa.py)."""
import numpy as np

from oracle import quip_oracle as O

BYTE_OF_POS = O.E8P_BYTE_OF_POS


def t2n_table():
    t = np.zeros(256, dtype=np.uint32)
    for s in range(256):
        par = bin(s).count("1") & 1
        sv = s ^ par
        e = 0
        for p in range(8):
            nib = (0xE if (sv >> (7 - BYTE_OF_POS[p])) & 1 else 0) ^ par
            e |= nib << (8 * (p & 3) + 4 * (p >> 2))
        t[s] = e
    return t


def t1n_table():
    packed = O.e8p_grid_packed_abs().astype(np.int64).view(np.uint64)
    t = np.zeros(256, dtype=np.uint32)
    for a in range(256):
        b = [(int(packed[a]) >> (8 * i)) & 0xFF for i in range(8)]
        nat = [b[BYTE_OF_POS[p]] for p in range(8)]                # natural position order (t1n_entry's two v_perm_b32)
        lo = sum(nat[p] << (8 * p) for p in range(4))
        hi = sum(nat[4 + p] << (8 * p) for p in range(4))
        t[a] = ((((lo >> 1) & 0x0F0F0F0F) ^ 0x08080808) | (((hi >> 1) & 0x0F0F0F0F) << 4)) & 0xFFFFFFFF
    return t


def test_nibble_tables_decode_every_code():
    t1, t2 = t1n_table(), t2n_table()
    codes = np.arange(1 << 16, dtype=np.uint32)
    u = t1[codes >> 8] ^ t2[codes & 0xFF]
    ref = O.e8p_decode_i8(codes.astype(np.uint16)).astype(np.int32)         # (65536, 8) = 4 w
    for p in range(8):
        nib = (u >> (8 * (p & 3) + 4 * (p >> 2))) & 0xF
        if p < 4:       # low nibble: v + 8, unsigned
            v = nib.astype(np.int32) - 8
        else:           # high nibble: v, two's complement
            v = ((nib.astype(np.int32) ^ 8) - 8)
        assert np.array_equal(2 * v + 1, ref[:, p]), p


def test_two_mfma_identity_gives_eight_times_the_digit_sums():
    rng = np.random.default_rng(5)
    t1, t2 = t1n_table(), t2n_table()
    ncodes = 512                                       # a row of 4096 weights
    codes = rng.integers(0, 1 << 16, size=(16, ncodes), dtype=np.uint32)
    codes[0, :256] = np.arange(256) << 8               # every abs entry, every sign byte at least once
    codes[1, :256] = np.arange(256)
    w4 = O.e8p_decode_i8(codes.astype(np.uint16)).astype(np.int64).reshape(16, -1)      # (16, 4096)
    dig = rng.integers(-128, 128, size=(3, ncodes * 8), dtype=np.int64)                 # three balanced digit planes
    dig[:, :8] = -128
    want = 8 * (dig @ w4.T)                                                              # (3, 16)
    u = t1[codes >> 8] ^ t2[codes & 0xFF]                                                # (16, ncodes) dwords
    raw = u.astype("<u4").view(np.uint8).reshape(16, ncodes, 4).view(np.int8).astype(np.int64)      # int8 as the MFMA reads them
    msk = (u & 0x0F0F0F0F).astype("<u4").view(np.uint8).reshape(16, ncodes, 4).astype(np.int64)
    d = dig.reshape(3, ncodes, 8)
    xlo, xhi = d[:, :, :4], d[:, :, 4:]                # positions 0..3 -> low nibbles, 4..7 -> high nibbles of bytes 0..3
    r1 = np.einsum("pcb,ncb->pn", xhi, raw)
    r2hi = np.einsum("pcb,ncb->pn", xhi, msk)
    r2lo = np.einsum("pcb,ncb->pn", xlo, msk)
    sxhi, sxlo = xhi.sum(axis=(1, 2)), xlo.sum(axis=(1, 2))
    got = (r1 - r2hi) + 16 * r2lo + (8 * sxhi - 120 * sxlo)[:, None]
    assert np.array_equal(got, want)
    assert np.abs(got).max() < 2 ** 31


def test_half_plane_layout_is_a_bijection():
    """nib_half_of / nib_byte_of (e8p_gemv_core.hip.h): digit k of a plane -> (half, byte) covers both halves exactly once and
    keeps the four digits of positions 0..3 (4..7) of an 8-group in one dword"""
    k = np.arange(8192)
    half, byte = (k >> 2) & 1, ((k >> 3) << 2) | (k & 3)
    assert len(set(zip(half.tolist(), byte.tolist()))) == 8192
    assert byte.max() == 4095
    assert np.array_equal(byte[half == 0].reshape(-1, 4)[:, 0] % 4, np.zeros(1024, dtype=np.int64))
