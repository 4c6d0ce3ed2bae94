"""Exhaustive code-space parity of every HIP decode path (SURVEY section 4's pyramid: "decode of all 65 536 codes",
codebook/e8p12.py:82-103, origin_order.cu:211-253 -- so far only the CPU oracle had it).

For each codebook a code matrix is built that holds EVERY code value (all 65 536 E8P12 codes; every 16-bit main and every
16-bit residual of E8P12RVQ4B; every main code and every E81B index of E8P12RVQ3B; every D4 byte in every column; every byte of
an HI code word, i.e. every nibble pair), and every weight of it is read back through each product path by multiplying with
unit vectors -- the result of a product with e_j is column j of the dense matrix, one exactly representable weight per output,
so the comparison with the oracle's decoded matrix is BIT FOR BIT (fp16 patterns), not a tolerance:

  decompress_*_origorder | the reference ops *_mm_origorder at M = 1 and M = 16 | the bs = 1 matrix-core GEMV on digit planes
  (the production transform launch makes the planes of e_j: its input is H e_j, a +-1 vector) | rows mode (several rows per pass
  over the codes) | the single-pass skinny kernel | the fused dequant + MFMA tile GEMM (M = 512 rows: the identity matrix).

The two RVQ codebooks round main + s * resid once to fp16 per weight in the reference (origin_order.cu:330-385); the integer
paths sum it exactly and round the OUTPUT once, which for a unit vector is the same single rounding."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import quip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K = 512


def _cb(cbid, scale=None):
    import quip_for_all_amd as Q
    kw = {} if scale is None else {"opt_resid_scale": scale}
    return Q.codebook.codebook_id[cbid](inference=True, **kw).to(DEV)


def _codes(cbid):
    """a code matrix (n, ...) that holds every code value of the codebook, with k = 512 weights per row"""
    if cbid == "E8P12":                       # 1024 rows x 64 codes: each of the 65 536 codes exactly once
        return np.arange(65536, dtype=np.uint32).reshape(1024, 64).astype(np.uint16).view(np.int16)
    if cbid == "E8P12RVQ4B":                  # main = every code; resid = an odd multiple of it mod 2^16: every code too
        main = np.arange(65536, dtype=np.uint32).reshape(1024, 64)
        resid = (main * 40503 + 12345) & 0xFFFF
        assert len(np.unique(resid)) == 65536
        return ((main << 16) | resid).astype(np.uint32).view(np.int32)
    if cbid == "E8P12RVQ3B":                  # main = every code, resid = every E81B index 256 times, against varying mains
        main = np.arange(65536, dtype=np.int64).reshape(1024, 64)
        resid = (main * 197 + 13) & 0xFF
        assert all(len(np.unique(resid[main >> 8 == a])) == 256 for a in (0, 17, 255))
        return O.rvq3_pack(((main << 8) + resid).astype(np.int32))
    if cbid == "D4":                          # 256 rows x 128 one-byte codes: every byte value in every column
        r, j = np.meshgrid(np.arange(256), np.arange(128), indexing="ij")
        q = ((r + 37 * j) & 0xFF).astype(np.uint8)
        assert all(len(np.unique(q[:, c])) == 256 for c in range(128))
        return q
    if cbid == "HI":                          # 256 rows x 64 words: every value of every byte of a word (every nibble pair)
        r, j, b = np.meshgrid(np.arange(256), np.arange(64), np.arange(4), indexing="ij")
        by = ((r * (2 * b + 1) + 37 * j + 11 * b) & 0xFF).astype(np.uint32)
        assert all(len(np.unique(by[:, c, bb])) == 256 for c in (0, 63) for bb in range(4))
        return (by[..., 0] | (by[..., 1] << 8) | (by[..., 2] << 16) | (by[..., 3] << 24)).astype(np.uint32).view(np.int32)
    raise KeyError(cbid)


def _hadamard_rows(k):
    i = np.arange(k)
    pc = np.zeros((k, k), dtype=np.int64)
    x = i[:, None] & i[None, :]
    while x.any():
        pc += x & 1
        x >>= 1
    return np.where(pc & 1, -1.0, 1.0).astype(np.float16)          # H[j] = H e_j (symmetric)


CASES = [("E8P12", None), ("E8P12RVQ4B", 1 / 3.45), ("E8P12RVQ4B", -1.0), ("E8P12RVQ3B", 1 / 2.04), ("D4", None), ("HI", None)]


@pytest.mark.parametrize("cbid,scale", CASES)
def test_every_code_through_every_decode_path(cbid, scale):
    cb = _cb(cbid, scale)
    q = _codes(cbid)
    ref = O.decompress(cbid, q, 0.0 if scale is None else scale)           # (n, 512) fp16: the oracle's decode
    n = ref.shape[0]
    assert ref.shape[1] == K
    refT = np.ascontiguousarray(ref.T)                                      # row j = what a product with e_j must return
    Qd = torch.from_numpy(q).to(DEV)
    eye = torch.eye(K, dtype=torch.float16, device=DEV)
    rs = float(getattr(cb, "planes_resid_scale", 0.0))

    def same(got, want, what):
        g = got.detach().cpu().numpy()
        assert g.dtype == np.float16 and g.shape == want.shape, (what, g.shape, want.shape)
        bad = (g.view(np.uint16) != want.view(np.uint16)) & ~((g == 0) & (want == 0))       # (+0 == -0)
        assert not bad.any(), f"{cbid} {what}: {int(bad.sum())} of {bad.size} weights differ, first at {np.argwhere(bad)[0]}"

    # 1. dense decode
    same(cb.decompress_weight(Qd), ref, "decompress")
    # 2. the reference ops at M = 1 and M = 16
    same(torch.cat([cb.mm(eye[j:j + 1], Qd) for j in range(0, K, 7)]), refT[0:K:7], "mm_origorder M=1")
    same(torch.cat([cb.mm(eye[j:j + 16], Qd) for j in range(0, K, 16)]), refT, "mm_origorder M=16")
    # 3. the bs = 1 matrix-core GEMV on the digit planes of e_j, made by the production transform launch from H e_j
    H = torch.from_numpy(_hadamard_rows(K)).to(DEV)
    rows = []
    for j in range(K):
        planes = torch.ops.quip_lib.had_transform_planes_fused(H[j:j + 1], K, 1, None, True, None, 1.0 / K, None, 1e-5, None, rs)
        rows.append(cb.mm_planes(planes, Qd))
    same(torch.cat(rows), refT, "gemv on digit planes (M = 1)")
    # 4. rows mode: several rows per pass over the codes
    rows = []
    for j in range(0, K, 16):
        planes = torch.ops.quip_lib.had_transform_planes_rows(H[j:j + 16].contiguous(), K, 1, None, True, None, 1.0 / K, None, 1e-5, None, rs)
        rows.append(cb.mm_planes_rows(planes, Qd))
    same(torch.cat(rows), refT, "rows mode (M = 16)")
    # 5. the single-pass skinny kernel (fp16 MFMA on decoded fp16 weights)
    same(torch.cat([cb.mm_skinny(eye[j:j + 32].contiguous(), Qd) for j in range(0, K, 32)]), refT, "skinny (M = 32)")
    # 6. the fused dequant + MFMA tile GEMM
    batched = cb.mm_batched(eye, Qd) if hasattr(cb, "mm_batched") else torch.ops.quip_lib.e8p_mm_batched(eye, Qd, cb.grid_packed_abs)
    same(batched, refT, "tile GEMM (M = 512)")
    # the codes covered everything there is to cover
    if cbid == "E8P12":
        assert len(np.unique(q.view(np.uint16))) == 65536


@pytest.mark.parametrize("mode", [4, 16, 24, 32])
def test_decode_core_of_every_table_mode_decodes_every_code(mode):
    """the persistent launches take their LDS tables, look-up addresses and B fragments from the shared decode
    core (csrc/e8p_gemv_core.hip.h) and are compared with the stage-wise step / the float64 model only to a few fp16 ulps -- one
    mis-decoded code in 65 536 would not show there.  quip_e8p_decode_probe (csrc/decode_probe.hip) runs that core -- table
    build, item_addresses<mode>, the fragments fed to v_mfma_i32_16x16x64_i8 -- on every code once and reads the fragments back
    THROUGH the matrix core (one-hot A rows): bit for bit the oracle's 4 w, in the nibble mode of round 6 (4: what the three
    launches ship) and the byte-table modes of rounds 1-5 (16 / 16, 32 / 16, 32 / 32 copies)."""
    from quip_for_all_amd import capi
    L = capi.lib()
    ids = np.arange(1 << 16, dtype=np.uint32)
    code_of = ((ids * 40503 + 12345) & 0xFFFF).astype(np.uint16)          # (a bijection: odd multiplier)
    assert len(np.unique(code_of)) == 65536
    # tile t, row n, code j of the row (id = (16 t + n) 64 + j) -> lane order: [t][c][q][n][i], j = 32 c + 8 q + i
    tiles = code_of.reshape(64, 16, 2, 4, 8).transpose(0, 2, 3, 1, 4).copy()
    codes = torch.from_numpy(tiles.view(np.int16).reshape(-1)).to(DEV)
    grid = torch.from_numpy(O.e8p_grid_packed_abs()).to(DEV)
    out = torch.full((1 << 16, 8), 99, dtype=torch.int8, device=DEV)
    L.quip_e8p_decode_probe.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int32, ctypes.c_void_p]
    L.quip_e8p_decode_probe.restype = ctypes.c_int
    rc = L.quip_e8p_decode_probe(grid.data_ptr(), codes.data_ptr(), out.data_ptr(), mode, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    want = O.e8p_decode_i8(code_of)
    got = out.cpu().numpy()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, (mode, len(bad), [(hex(int(code_of[i])), got[i].tolist(), want[i].tolist()) for i in bad[:4]])


def test_block_engine_products_on_a_model_that_holds_every_code():
    """the persistent block launch decodes with the same core (e8p_gemv_core.hip.h) on its own request / slot machinery:
    a 7B-shaped block whose seven code matrices hold every E8P12 code (a permuted ramp) against the stage-wise step, whose
    GEMV the test above pins code by code.  Through round 4 the two agreed bit for bit; since round 5 the launch's edges add in
    another order and round its planes against the norm bound (decode_block.hip: edge), so the products see inputs that differ
    in the last of 18-22 bits: logits within 2 (4 sqrt(1) + 2) = 12 fp16 ulps of rms(logits) (tests/test_gpu_block_engine.py),
    cache rows within 2^-8 of their maximum -- a mis-decoded code (any of the 65 536, 8 weights wrong by >= 1/2) in any of the
    seven matrices moves the logits by hundreds of ulps."""
    import os
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=4096, ffn=11008, layers=1, heads=32, kv_heads=32, vocab=2048)

    def make(engine):
        old = os.environ.get("QUIP_BLOCK_ENGINE")
        os.environ["QUIP_BLOCK_ENGINE"] = "1" if engine else "0"
        try:
            return D.LlamaDecoder(shape, "E8P12", max_len=16, device=DEV, seed=3, device_init=True)
        finally:
            if old is None:
                os.environ.pop("QUIP_BLOCK_ENGINE", None)
            else:
                os.environ["QUIP_BLOCK_ENGINE"] = old
    a, b = make(True), make(False)
    with torch.no_grad():
        for La, Lb in zip(a.layers, b.layers):
            for i, k in enumerate(("q", "k", "v", "o", "gate", "up", "down")):
                m = La[k]
                ramp = (torch.arange(m.Qidxs.numel(), device=DEV, dtype=torch.int64) * (2 * i + 3) + 977 * i) & 0xFFFF
                code = (ramp - 65536 * (ramp >= 32768)).to(torch.int16).reshape(m.Qidxs.shape)
                assert len(torch.unique(code)) == 65536
                m.Qidxs.copy_(code)
                Lb[k].Qidxs.copy_(code)
                for name in ("had_left", "had_right"):
                    if getattr(m, name) is not None:
                        getattr(Lb[k], name).copy_(getattr(m, name))
    for dec in (a, b):
        dec.reset(first_token=7)              # (reset() rebuilds the engine descriptors: the modules were edited)
    assert a.block_eng and not b.block_eng and b.ffn_eng
    with torch.no_grad():
        for t in range(3):
            la, lb = a.step().clone(), b.step().clone()
            assert a.engine_status() == 0
            rms = lb.float().pow(2).mean().sqrt().item()
            ulps = (la.float() - lb.float()).abs().max().item() / 2.0 ** (np.floor(np.log2(rms)) - 10)
            assert ulps <= 12.0, (t, ulps)
            a.tok.copy_(b.tok)
    for ca, cb in ((a.kcache, b.kcache), (a.vcache, b.vcache)):
        assert (ca[:, :, :3].float() - cb[:, :, :3].float()).abs().max().item() <= 2.0 ** -8 * cb[:, :, :3].float().abs().max().item()
