"""Constant tables of the five codebooks, generated (vectorised numpy) rather
than stored.  Formats follow SURVEY.md appendix A; the generating rules restate
codebook/e8p12.py:28-79, e8p12_rvq3.py:16-62, d4.py:26-96, hi.py:41-50 of the
reference.  tests/test_tables.py checks every table against the golden
fixtures and SHA-256 anchors produced by the reference's own code."""
from functools import lru_cache

import numpy as np

_NORM12 = np.array([
    [3, 1, 1, 1, 3, 3, 3, 3], [1, 3, 1, 1, 3, 3, 3, 3], [1, 1, 3, 1, 3, 3, 3, 3],
    [1, 1, 1, 3, 3, 3, 3, 3], [3, 3, 3, 1, 3, 3, 1, 1], [3, 3, 3, 1, 3, 1, 3, 1],
    [3, 3, 3, 1, 1, 3, 3, 1], [3, 3, 3, 1, 3, 1, 1, 3], [3, 3, 3, 1, 1, 3, 1, 3],
    [3, 3, 3, 1, 1, 1, 3, 3], [3, 3, 1, 3, 3, 3, 1, 1], [3, 3, 1, 3, 3, 1, 3, 1],
    [3, 3, 1, 3, 1, 3, 3, 1], [3, 3, 1, 3, 3, 1, 1, 3], [3, 3, 1, 3, 1, 3, 1, 3],
    [3, 3, 1, 3, 1, 1, 3, 3], [3, 1, 3, 3, 3, 3, 1, 1], [3, 1, 3, 3, 3, 1, 3, 1],
    [3, 1, 3, 3, 1, 3, 3, 1], [3, 1, 3, 3, 3, 1, 1, 3], [3, 1, 3, 3, 1, 3, 1, 3],
    [1, 3, 3, 3, 1, 1, 3, 3], [1, 3, 3, 3, 3, 3, 1, 1], [1, 3, 3, 3, 3, 1, 3, 1],
    [1, 3, 3, 3, 1, 3, 3, 1], [1, 3, 3, 3, 3, 1, 1, 3], [1, 3, 3, 3, 1, 3, 1, 3],
    [1, 1, 3, 3, 1, 3, 3, 3], [3, 3, 1, 1, 3, 3, 3, 1]], dtype=np.int64)  # = 2 * value

_E8P_PERM = [0, 2, 1, 3, 4, 6, 5, 7]


@lru_cache(maxsize=None)
def e8p_grid_packed_abs() -> np.ndarray:
    """int64[256]; byte j (LE) = int8(4 * |a|) of column _E8P_PERM[j], byte 7 negated
    when the row's coordinate sum is odd."""
    # all abs patterns over {1,3,5,7}/2 as "twice the value" integers; lexicographic
    # order of the 8-digit base-4 counter == lexicographic order of the rows
    digits = (np.arange(4 ** 8)[:, None] >> (2 * np.arange(7, -1, -1))) & 3
    twice = 2 * digits + 1                               # 1,3,5,7
    keep = (twice ** 2).sum(1) <= 40                     # ||.||^2 <= 10
    rows = np.concatenate([twice[keep], _NORM12], axis=0)
    assert rows.shape == (256, 8)
    b = 2 * rows[:, _E8P_PERM]                           # 4 * value
    odd = (rows.sum(1) // 2) % 2 == 1                    # coordinate sum odd
    b[odd, 7] = -b[odd, 7]
    return np.ascontiguousarray(b.astype(np.int8)).view(np.int64).reshape(256)


@lru_cache(maxsize=None)
def e8p_full_grid() -> np.ndarray:
    """float32[65536, 8]: row c = the 8 weights of E8P12 code c (inference=False only)."""
    tab = e8p_grid_packed_abs().view(np.int8).reshape(256, 8).astype(np.int32)
    c = np.arange(1 << 16)
    s, a = c & 255, c >> 8
    par = np.bitwise_xor.reduce((s[:, None] >> np.arange(8)) & 1, axis=1)
    sv = s ^ par
    byte_of_pos = np.array(_E8P_PERM)
    mag = tab[a][:, byte_of_pos]
    neg = (sv[:, None] >> (7 - byte_of_pos)[None, :]) & 1
    w4 = np.where(neg == 1, -mag, mag) + np.where(par == 1, -1, 1)[:, None]
    return (w4 / 4.0).astype(np.float32)


@lru_cache(maxsize=None)
def e81b_grid() -> np.ndarray:
    """float32[256, 8] residual codebook of E8P12RVQ3B."""
    def lex(points):
        order = np.lexsort(points.T[::-1])
        return points[order]
    t = (np.arange(3 ** 8)[:, None] // 3 ** np.arange(7, -1, -1)) % 3 - 1     # {-1,0,1}^8
    t = t[((t ** 2).sum(1) <= 2) & (t.sum(1) % 2 == 0)]
    h = ((np.arange(256)[:, None] >> np.arange(7, -1, -1)) & 1) - 0.5          # {-1/2,1/2}^8
    h = h[h.sum(1) % 2 == 0]
    plus = 2.0 * np.eye(8)
    minus = -2.0 * np.eye(8)[:7]
    g = np.concatenate([lex(t.astype(np.float64)), lex(h), plus, minus], axis=0)
    assert g.shape == (256, 8)
    return g.astype(np.float32)


@lru_cache(maxsize=None)
def e81b_grid_packed() -> np.ndarray:
    """int32[256]: nibble i = int4(2 * value) of column [0,2,4,6,1,3,5,7][i]."""
    v = (2 * e81b_grid()[:, [0, 2, 4, 6, 1, 3, 5, 7]]).astype(np.int64) & 0xF
    return (v << (4 * np.arange(8))).sum(1).astype(np.uint32).view(np.int32)


@lru_cache(maxsize=None)
def d4_grid() -> np.ndarray:
    """float32[256, 4] deep-hole-centred D4 codebook."""
    h, o, t = 0.5, 1.5, 2.5
    base = np.empty((32, 4))
    base[0], base[1] = h, o
    for lo in range(2, 8):
        fill, spot = (h, o) if lo & 1 else (o, h)
        base[lo] = fill
        base[lo, 0] = base[lo, lo >> 1] = spot
    for lo in range(8, 12):
        base[lo] = h
        base[lo, lo & 3] = o
    for lo in range(12, 16):
        base[lo] = o
        base[lo, lo & 3] = h
    for lo in range(16, 20):
        base[lo] = h
        base[lo, lo & 3] = t
    for lo in range(20, 32):
        i4, i3 = (lo - 20) & 3, (lo - 20) >> 2
        i3 += i3 >= i4
        base[lo] = h
        base[lo, i4], base[lo, i3] = o, t
    g = np.tile(base, (8, 1))                       # code = flags << 5 | lo
    flags = np.repeat(np.arange(8), 32)
    g[(flags & 1) == 1, 2] *= -1                     # bit 5
    g[(flags & 2) == 2, 1] *= -1                     # bit 6
    odd = g.sum(1) % 2 != 0
    g[odd, 3] *= -1                                  # make the coordinate sum even
    g[(flags & 4) == 4] *= -1                        # bit 7: negate all
    return g.astype(np.float32)


HI_NIBBLE_COLS = (0, 2, 4, 6, 1, 3, 5, 7)
