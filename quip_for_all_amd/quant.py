"""Hadamard helpers with the reference's names and argument meaning
(quant.py:8-39, 72-88): get_hadK, matmul_hadU_cuda, matmul_hadUt_cuda."""
import math
import os

import torch

_HAD_TABLES = None


def _had_tables():
    """Non-power-of-two Hadamard factors for use_rand=False (quant.py:8,34-39).
    The reference ships them as hadamard.safetensors; this build reads the same
    file when QUIP_HADAMARD_TABLES points at it."""
    global _HAD_TABLES
    if _HAD_TABLES is None:
        path = os.environ.get("QUIP_HADAMARD_TABLES", "")
        if path and os.path.exists(path):
            from safetensors.torch import load_file
            _HAD_TABLES = load_file(path)
        else:
            _HAD_TABLES = {}
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
