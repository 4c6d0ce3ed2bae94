"""The compile boundary of the reference (register_lib.py:22-192: every op has an abstract implementation so that
`torch.compile(mode="reduce-overhead", fullgraph=True)` traces the decode step, example_generate.py:68-70):
torch.library.opcheck on the 11 reference ops (schema + fake implementation vs the real one: shapes, dtypes, strides,
devices) and a full-graph trace of QuantLinear.forward for every codebook at M = 1 and M = 40 with the `aot_eager`
backend, compared with eager.  No Inductor / Triton is involved: this only proves that a reference user's
torch.compile call traces through the ops."""
import numpy as np
import pytest
import torch

from oracle import quip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OPCHECKS = ("test_schema", "test_faketensor")


def _layer(cb, fin=256, fout=512, seed=1):
    import quip_for_all_amd as Q
    P = O.make_layer(cb, fin, fout, seed=seed)
    return Q.qlinear.QuantLinear.from_params(P).to(DEV).eval()


def _x(m, k, seed=0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal((m, k)).astype(np.float16)).to(DEV)


def test_opcheck_reference_ops():
    import quip_for_all_amd  # noqa: F401
    ops = torch.ops.quip_lib
    x = _x(3, 256)
    torch.library.opcheck(ops.hadamard.default, (x, 0.25), test_utils=OPCHECKS)
    torch.library.opcheck(ops.hadamard.default, (x.float(), 1.0), test_utils=OPCHECKS)
    for cb, mm, dec in (("E8P12", ops.e8p_mm_origorder, ops.decompress_e8p_origorder),
                        ("E8P12RVQ3B", ops.e8prvq3_mm_origorder, ops.decompress_e8prvq3_origorder),
                        ("E8P12RVQ4B", ops.e8prvq4_mm_origorder, ops.decompress_e8prvq4_origorder),
                        ("D4", ops.d4_mm_origorder, ops.decompress_d4_origorder),
                        ("HI", ops.hi_mm_origorder, ops.decompress_hi_origorder)):
        layer = _layer(cb)
        c = layer.codebook
        q = layer.Qidxs
        if cb == "E8P12":
            extra_mm, extra_dec = (c.grid_packed_abs,), (c.grid_packed_abs,)
        elif cb == "E8P12RVQ3B":
            extra_mm = extra_dec = (c.grid_packed_abs, c.e81b_grid_packed, float(c.opt_resid_scale))
        elif cb == "E8P12RVQ4B":
            extra_mm = extra_dec = (c.grid_packed_abs, float(c.opt_resid_scale))
        elif cb == "D4":
            extra_mm = extra_dec = (c.grid,)
        else:
            extra_mm = extra_dec = ()
        for m in (1, 5):
            torch.library.opcheck(mm.default, (_x(m, layer.q_in_features), q) + extra_mm, test_utils=OPCHECKS)
        torch.library.opcheck(dec.default, (q,) + extra_dec, test_utils=OPCHECKS)


@pytest.mark.parametrize("cb", ["E8P12", "E8P12RVQ3B", "E8P12RVQ4B", "D4", "HI"])
@pytest.mark.parametrize("m", [1, 40])
def test_quantlinear_forward_traces_as_one_graph(cb, m):
    layer = _layer(cb, 256, 512, seed=3)
    x = _x(m, 256, seed=m)
    with torch.no_grad():
        want = layer(x)
        torch._dynamo.reset()
        compiled = torch.compile(layer, backend="aot_eager", fullgraph=True)
        got = compiled(x)
        got2 = compiled(x)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert torch.equal(got, want) and torch.equal(got2, want)
