"""ctypes wrapper of oracle/quip_oracle.c (TEST INFRASTRUCTURE ONLY: checker + cpu_baseline)."""
import ctypes
import os
import subprocess
import time

import numpy as np

from . import quip_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libquip_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "quip_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = ctypes.CDLL(_SO)
        _lib.quip_oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(lib().quip_oracle_num_threads())


def decompress_e8p(qidxs, grid):
    q = np.ascontiguousarray(qidxs).view(np.uint16)
    w = np.empty((q.shape[0], q.shape[1] * 8), np.float32)
    lib().quip_oracle_decompress_e8p(_p(q), _p(np.ascontiguousarray(grid)), _p(w), ctypes.c_long(q.shape[0]),
                                     ctypes.c_long(q.shape[1]))
    return w


def fwht(x):
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().quip_oracle_fwht(_p(y), ctypes.c_long(y.shape[-1]))
    return y


def _prepare(P):
    """C argument buffers of a layer (built once; keeps the numpy arrays alive)"""
    assert P.codebook == "E8P12" and not P.per_channel
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    return [f32(P.SU), f32(P.SV), f32(P.bias), f32(P.had_left), f32(P.had_right),
            np.ascontiguousarray(P.Qidxs).view(np.uint16), np.ascontiguousarray(O.e8p_grid_packed_abs())]


def _call(P, keep, xx, y):
    c = ctypes
    lib().quip_oracle_qlinear_e8p(_p(xx), _p(y), c.c_long(P.in_features), c.c_long(P.out_features),
                                  c.c_long(P.q_in), c.c_long(P.q_out), _p(keep[5]), _p(keep[6]), _p(keep[0]),
                                  _p(keep[1]), _p(keep[2]), c.c_float(P.wscale_float), c.c_long(P.K_left),
                                  _p(keep[3]), c.c_long(P.K_right), _p(keep[4]))


def qlinear_forward(P, x):
    """one token row through the C forward; P is an oracle QLinearParams (E8P12, scalar Wscale)"""
    xx = np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.float32)
    y = np.empty(P.out_features, np.float32)
    _call(P, _prepare(P), xx, y)
    return y


def time_qlinear_forward(codebook, fin, fout, min_time=3.0):
    """seconds per call of the C forward alone (argument buffers and tables prepared outside the loop)"""
    P = O.make_layer(codebook, fin, fout, seed=1)
    xx = np.random.default_rng(0).standard_normal(fin).astype(np.float32)
    y = np.empty(P.out_features, np.float32)
    keep = _prepare(P)
    _call(P, keep, xx, y)
    n, t0 = 0, time.perf_counter()
    while True:
        _call(P, keep, xx, y)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_time and n >= 3:
            return dt / n


def time_dense_gemv(n, k, min_time=2.0):
    rng = np.random.default_rng(0)
    w = rng.standard_normal((n, k)).astype(np.float32)
    x = rng.standard_normal(k).astype(np.float32)
    y = np.empty(n, np.float32)
    args = (_p(w), _p(x), _p(y), ctypes.c_long(n), ctypes.c_long(k))
    lib().quip_oracle_dense_gemv(*args)
    cnt, t0 = 0, time.perf_counter()
    while True:
        lib().quip_oracle_dense_gemv(*args)
        cnt += 1
        dt = time.perf_counter() - t0
        if dt >= min_time and cnt >= 3:
            return dt / cnt
