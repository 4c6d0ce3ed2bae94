"""Hadamard helpers with the reference's names and argument meaning
(quant.py:8-39, 72-88): get_hadK, matmul_hadU_cuda, matmul_hadUt_cuda."""
import math
import os

import torch

_HAD_TABLES = None
_HAD_TABLES_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "hadamard_tables.npz")


def _had_tables():
    """Non-power-of-two Hadamard factors for use_rand=False (quant.py:8,34-39): the +-1 matrices of order 4 * odd
    (12 .. 252) the reference ships as the data file hadamard.safetensors, bundled bit-packed in
    data/hadamard_tables.npz (made by tests/golden/make_hadamard_tables.py; every matrix satisfies H H^T = n I,
    checked on CPU by tests/test_hadamard_tables.py).  QUIP_HADAMARD_TABLES may point at a safetensors file with
    the reference's layout instead.  A missing file raises: silently padding to the next power of two would change
    (K, padded n) and with them the shape of Qidxs of every checkpoint quantised with use_rand=False."""
    global _HAD_TABLES
    if _HAD_TABLES is None:
        path = os.environ.get("QUIP_HADAMARD_TABLES", "")
        if path:
            if not os.path.exists(path):
                raise FileNotFoundError(f"QUIP_HADAMARD_TABLES={path} does not exist")
            from safetensors.torch import load_file
            _HAD_TABLES = {k: v.to(torch.float32) for k, v in load_file(path).items()}
        else:
            if not os.path.exists(_HAD_TABLES_FILE):
                raise FileNotFoundError(
                    f"{_HAD_TABLES_FILE} is missing: get_hadK(use_rand=False) needs the Hadamard factor tables "
                    "(regenerate with tests/golden/make_hadamard_tables.py or set QUIP_HADAMARD_TABLES)")
            import numpy as np
            z = np.load(_HAD_TABLES_FILE)
            tabs = {}
            for n in z["orders"].tolist():
                bits = np.unpackbits(z[f"bits_{n}"])[: n * n].reshape(n, n)
                tabs[str(n)] = torch.from_numpy(bits.astype(np.float32) * 2.0 - 1.0)
            _HAD_TABLES = tabs
    return _HAD_TABLES


def next_power_of_2(n):
    return 1 if n == 0 else 2 ** math.ceil(math.log(n, 2))


def get_power_of_2(n):
    """(e, base) with n = 2**e * base, base odd (quant.py:17-23)."""
    k = 0
    while n % 2 == 0:
        n //= 2
        k += 1
    return k, n


def get_hadK(n, use_rand=True):
    """(hadK, K, padded_n) exactly as quant.py:26-39."""
    exp, base = get_power_of_2(n)
    if base == 1:
        return None, 1, n
    if use_rand:
        import scipy.stats
        rand_mat = torch.tensor(scipy.stats.special_ortho_group.rvs(base)).to(torch.float32)
        return rand_mat, base, n
    pad_n = next_power_of_2(n)
    tables = _had_tables()
    if exp < 2 or str(base * 4) not in tables:
        return None, 1, pad_n
    return tables[str(base * 4)] / math.sqrt(base * 4), base * 4, n


def matmul_hadU_cuda(X, hadK, K, n, scale=None, transpose=False):
    """(hadK (x) H_{n/K}) X / sqrt(n/K) * scale on the (.., K, n/K) view; one fused
    launch instead of pad + hadamard + hadK@ (quant.py:72-84)."""
    lead = X.shape[:-1]
    x2 = X.reshape(-1, X.shape[-1])
    s = (1.0 if scale is None else scale) / math.sqrt(n // K)
    had = None
    if K > 1:
        had = hadK.to(device=X.device, dtype=torch.float16).contiguous()
    y = torch.ops.quip_lib.had_transform(x2, n, n, K, had, transpose, None, None, None, None, s)
    return y.reshape(*lead, n)


def matmul_hadUt_cuda(X, hadK, K, n, scale=None):
    return matmul_hadU_cuda(X, hadK, K, n, scale=scale, transpose=True)


# ---- quantise-time helpers (any float dtype, any device: plain torch; quant.py:42-65, 90-135) -------

def _fwht_lastdim(y):
    """unnormalised Walsh-Hadamard transform (Sylvester order) of the last dimension, length 2^e"""
    L = y.shape[-1]
    lead = y.shape[:-1]
    h = 1
    while h < L:
        y = y.reshape(*lead, L // (2 * h), 2, h)
        y = torch.stack((y[..., 0, :] + y[..., 1, :], y[..., 0, :] - y[..., 1, :]), dim=-2)
        h *= 2
    return y.reshape(*lead, L)


def matmul_hadU(X, hadK, K, n, transpose=False):
    """X (.., in <= n) -> X_pad (hadK (x) H_{n/K})^T / sqrt(n/K) on the row-major (K, n/K) view of the last
    dimension: the torch statement of the transform the HIP kernels apply to fp16 activations, used at
    quantise time on fp32 / fp64 weights and Hessians (quant.py:42-65)."""
    if X.shape[-1] != n:
        X = torch.nn.functional.pad(X, (0, n - X.shape[-1]))
    L = n // K
    y = _fwht_lastdim(X.reshape(-1, K, L))
    if K > 1:
        hk = hadK.to(device=y.device, dtype=y.dtype)
        y = (hk.T if transpose else hk) @ y
    return (y / math.sqrt(L)).reshape(X.shape)


def matmul_hadUt(X, hadK, K, n):
    return matmul_hadU(X, hadK, K, n, transpose=True)


def block_LDL(L, b):
    """Cholesky factor L (n, n) -> block-unit-lower factor: every block column i (width b) is multiplied from
    the right by the inverse of its diagonal block, so the diagonal blocks become identities (quant.py:90-102)."""
    n = L.shape[0]
    assert n % b == 0
    m = n // b
    blocks = L.reshape(m, b, m, b)
    diag_inv = torch.linalg.inv(torch.stack([blocks[i, :, i, :] for i in range(m)]))      # (m, b, b)
    out = torch.einsum("nib,ibc->nic", L.reshape(n, m, b), diag_inv).reshape(n, n)
    if torch.isnan(out).any():
        raise ValueError("Hessian is not invertible")
    return out


def LDLQ(Wr, Hr, L, cb, quip_tune_iters=0, buf_cols=128):
    """Block LDL adaptive rounding (quant.py:105-135): going through the column groups of width cb.codesz
    from the right, group k is rounded to the codebook AFTER the rounding error of the groups to its right has
    been fed back through the block-LDL factor of the (incoherence-processed) Hessian,
        hatW_k = Q( W_k + (W_{>k} - hatW_{>k}) L_{>k,k} ),
    so that (W - hatW) L stays small.  The feedback of everything right of the current panel of `buf_cols`
    columns is one GEMM per panel (the reference's LDLQ_buffered, quant.py:138-230); `quip_tune_iters` extra
    sweeps re-round each group against the exact proxy gradient.  Returns (hatWr, Qidxs (m, n / codesz))."""
    m, n = Wr.shape
    b = cb.codesz
    assert n % b == 0
    Lb = block_LDL(L.clone(), b)
    hatWr = torch.zeros_like(Wr)
    Qidxs = torch.zeros(m, n // b, dtype=cb.idx_dtype, device=Wr.device)
    panel = max(b, (min(buf_cols, n) // b) * b)
    hi_p = n
    while hi_p > 0:
        lo_p = max(0, hi_p - panel)
        # error of the finished columns right of the panel, fed back into the whole panel at once
        base = (Wr[:, hi_p:] - hatWr[:, hi_p:]) @ Lb[hi_p:, lo_p:hi_p] if hi_p < n else 0.0
        for lo in range(hi_p - b, lo_p - 1, -b):
            hi = lo + b
            target = Wr[:, lo:hi] + (Wr[:, hi:hi_p] - hatWr[:, hi:hi_p]) @ Lb[hi:hi_p, lo:hi]
            if hi_p < n:
                target = target + base[:, lo - lo_p:hi - lo_p]
            vals, idx = cb.quantize(target)
            hatWr[:, lo:hi] = vals
            Qidxs[:, lo // b] = idx
        hi_p = lo_p
    for _ in range(quip_tune_iters):
        for lo in range(n - b, -1, -b):
            hi = lo + b
            target = hatWr[:, lo:hi] + (Wr - hatWr) @ Hr[:, lo:hi] @ torch.linalg.inv(Hr[lo:hi, lo:hi])
            vals, idx = cb.quantize(target)
            hatWr[:, lo:hi] = vals
            Qidxs[:, lo // b] = idx
    return hatWr, Qidxs


LDLQ_buffered = LDLQ
