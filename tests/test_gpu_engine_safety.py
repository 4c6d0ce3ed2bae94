"""The persistent launches spin across workgroups and need every workgroup resident.  What happens when that does not hold:
the launch is refused up front where it can be known (CU count, occupancy query: csrc/capi.hip persistent_grid_fits), and a
launch that finds the device shared at run time gives up on a bounded wait, answers NaN, remembers the position -- and
`LlamaDecoder.generate` decodes that token and the ones behind it again on the stage-wise step instead of raising."""
import ctypes
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _decoder(block_engine, ffn_engine, layers=2, max_len=40, seed=3):
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=4096, ffn=11008, layers=layers, heads=32, kv_heads=32, vocab=2048)
    old = {k: os.environ.get(k) for k in ("QUIP_BLOCK_ENGINE", "QUIP_FFN_ENGINE")}
    os.environ["QUIP_BLOCK_ENGINE"] = "1" if block_engine else "0"
    os.environ["QUIP_FFN_ENGINE"] = "1" if ffn_engine else "0"
    np.random.seed(1234 + seed)
    try:
        return D.LlamaDecoder(shape, "E8P12", max_len=max_len, device=DEV, seed=seed, device_init=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same_weights(dst, src):
    with torch.no_grad():
        for Ld, Ls in zip(dst.layers, src.layers):
            for k in ("gate", "up", "down"):
                for name in ("had_left", "had_right"):
                    if getattr(Ls[k], name) is not None:
                        getattr(Ld[k], name).copy_(getattr(Ls[k], name))


def test_generate_falls_back_to_the_stagewise_step_when_the_device_is_shared():
    """32 CUs held by another stream's kernel for longer than a hand-off waits: the persistent launch cannot have its 256
    workgroups resident, gives up (no hang), and generate() returns the tokens of the stage-wise step"""
    from quip_for_all_amd import capi
    plain = _decoder(False, False)
    dec = _decoder(True, True)
    _same_weights(dec, plain)
    assert dec.block_eng and not plain.block_eng and not plain.ffn_eng
    expected = plain.generate(12, first_token=5, use_graph=True).cpu().numpy()
    dec.capture()                                     # (descriptors rebuilt by reset(): the factors were edited)
    assert dec.block_eng and dec.engine_status() == 0
    sink = torch.zeros(4, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream(priority=-1)      # (its own hardware queue: streams of equal priority may share one, which would serialise the two kernels)
    # ~8 s of shader clocks at 2.1 GHz: longer than the ~2-4 s after which a wait gives up
    capi.check(capi.lib().quip_debug_occupy(32, 100 * 1024, ctypes.c_int64(17_000_000_000), sink.data_ptr(), side.cuda_stream),
               "quip_debug_occupy")
    with warnings.catch_warnings(record=True) as wr:
        warnings.simplefilter("always")
        got = dec.generate(12, first_token=5, use_graph=True).cpu().numpy()
    torch.cuda.synchronize()
    assert any("gave up" in str(w.message) for w in wr), [str(w.message) for w in wr]
    assert not dec.block_eng and not dec.ffn_eng and dec.engine_status() == 0
    np.testing.assert_array_equal(got, expected)
    # and the decoder keeps working (stage-wise) afterwards
    again = dec.generate(12, first_token=5, use_graph=True).cpu().numpy()
    np.testing.assert_array_equal(again, expected)


def test_a_launch_that_fails_mid_generation_behind_a_prompt_keeps_the_prompt_rows():
    """the retry after a persistent launch gave up must not warm a new capture up on the LIVE cache (capture() runs
    two steps at positions 0 and 1).  A failure is injected behind a batched prompt pass (engine words written by hand after
    the decode loop): the resumed decode runs eagerly from the failed position, cache rows of the prompt stay what the prompt
    pass wrote, and the tokens in front of the failed position are kept."""
    plain = _decoder(False, False)
    dec = _decoder(True, True)
    _same_weights(dec, plain)
    prompt = torch.tensor([5, 9, 2, 7, 11], device=DEV)
    n = 8
    expected = plain.generate(n, prompt=prompt, use_graph=True).cpu().numpy()
    rows_ref = [k[:, :4].clone() for k in plain.kcache]
    clean = dec.generate(n, prompt=prompt, use_graph=True).cpu().numpy()
    fail_at = (prompt.numel() - 1) + 3                           # absolute position of the fourth generated token
    state = {"armed": True}
    real_status, real_fail = dec.engine_status, dec.engine_fail_position

    def status():
        if state["armed"]:
            return 0x2001
        return real_status()

    def fail_position():
        if state["armed"]:
            state["armed"] = False
            return fail_at
        return real_fail()

    dec.engine_status, dec.engine_fail_position = status, fail_position
    with warnings.catch_warnings(record=True) as wr:
        warnings.simplefilter("always")
        got = dec.generate(n, prompt=prompt, use_graph=True).cpu().numpy()
    del dec.engine_status, dec.engine_fail_position
    assert any("gave up" in str(w.message) for w in wr)
    assert not dec.block_eng and dec.graph is None               # the next call captures the stage-wise step
    for k_ref, k in zip(rows_ref, dec.kcache):
        assert torch.equal(k[:, :4], k_ref), "prompt rows of the cache were overwritten by the retry"
    np.testing.assert_array_equal(got[:3], clean[:3])            # tokens in front of the failed position are kept
    if np.array_equal(clean, expected):                          # (the two paths agree on this model: then so does the mix)
        np.testing.assert_array_equal(got, expected)
    again = dec.generate(n, prompt=prompt, use_graph=True).cpu().numpy()
    np.testing.assert_array_equal(again, expected)


def test_a_failed_launch_answers_nan_and_remembers_the_position():
    from quip_for_all_amd import capi
    dec = _decoder(True, True, layers=1)
    dec.reset(first_token=3)
    dec.pos.fill_(5)
    sink = torch.zeros(4, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream(priority=-1)      # (its own hardware queue: streams of equal priority may share one, which would serialise the two kernels)
    capi.check(capi.lib().quip_debug_occupy(16, 100 * 1024, ctypes.c_int64(17_000_000_000), sink.data_ptr(), side.cuda_stream),
               "quip_debug_occupy")
    with torch.no_grad():
        logits = dec.step()
    torch.cuda.synchronize()
    assert dec.engine_status() != 0
    assert dec.engine_fail_position() == 5
    assert torch.isnan(logits).all()
    dec.engine_reset()
    assert dec.engine_status() == 0 and dec.engine_fail_position() is None


def test_launch_counter_wrap_asks_for_a_fresh_workspace_and_generation_continues():
    """the hand-off tag carries 22 bits of launch counter: two launches before the wrap the launch leaves code 0xE000, the
    next one answers NaN at once, and generate() zeroes the workspace and decodes on -- still on the persistent launch"""
    ref = _decoder(True, True)
    dec = _decoder(True, True)
    _same_weights(dec, ref)
    expected = ref.generate(10, first_token=5, use_graph=True).cpu().numpy()
    dec.capture()
    dec.eng_ws[:4].view(torch.int32).fill_((1 << 22) - 6)         # the generation word: a few launches before the wrap
    with warnings.catch_warnings(record=True) as wr:
        warnings.simplefilter("always")
        got = dec.generate(10, first_token=5, use_graph=True).cpu().numpy()
    assert not wr, [str(w.message) for w in wr]
    assert dec.block_eng and dec.engine_status() == 0
    assert int(dec.eng_ws[:4].view(torch.int32).item()) < 64      # a fresh workspace
    np.testing.assert_array_equal(got, expected)


def test_hf_fast_decode_wrapper_redoes_a_step_the_shared_device_spoiled():
    """hf_fast (what load_quantized_model installs): a 7B-shaped HF model decoding on a StaticCache through the persistent
    launch -- a step whose launch gave up is answered by the stage-wise step (a warning, no NaN logits), the tokens are those
    of the stage-wise run, and the wrapper stays on the stage-wise step"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from transformers import AutoModelForCausalLM, LlamaConfig
    from tests.test_quantizer_host import _fill_random
    from quip_for_all_amd import capi
    from quip_for_all_amd.quantizer import QuipQuantizer
    from quip_for_all_amd.hf_static import HFStaticDecoder
    from quip_for_all_amd.hf_fast import enable_fast_decode
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=2, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=2048, max_position_embeddings=64, tie_word_embeddings=False)
    torch.manual_seed(0)
    model = AutoModelForCausalLM.from_config(cfg, dtype=torch.float16)
    QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0).convert_model(model)
    _fill_random(model, seed=5)
    model = model.to(DEV).eval()
    enable_fast_decode(model)
    fd = model._quip_fast_decode
    ids = torch.tensor([[1, 17, 42, 99, 7, 250]], device=DEV)
    expected, _ = HFStaticDecoder(model, max_cache_len=64).generate(ids, 10, "eager")
    assert fd.dec is not None and fd.dec.block_eng and fd.fast_steps == 9
    h = HFStaticDecoder(model, max_cache_len=64)
    got = [int(h.prefill(ids))]
    nxt = h.decode_one_token(h.tok, h.pos)
    h.tok.copy_(nxt)
    h.pos += 1
    got.append(int(nxt))
    # a launch that gave up leaves its code in the workspace and every wait of the next launch ends at once (engine_sync.hip.h):
    # that state, set by hand (how a shared device produces it: the first test above; two streams of a long test session may
    # share a hardware queue, which serialises the kernels instead of letting them contend)
    fd.dec.eng_ws[4:8].view(torch.int32).fill_(0x5001)
    with warnings.catch_warnings(record=True) as wr:
        warnings.simplefilter("always")
        for _ in range(8):
            nxt = h.decode_one_token(h.tok, h.pos)
            h.tok.copy_(nxt)
            h.pos += 1
            got.append(int(nxt))
    torch.cuda.synchronize()
    assert any("gave up" in str(w.message) for w in wr), [str(w.message) for w in wr]
    assert not fd.dec.block_eng and fd.dec.engine_status() == 0
    got = torch.tensor(got, device=DEV)
    # every step behind the failure was answered by the stage-wise path: the same harness again (the wrapper now stage-wise)
    # decodes the same tokens; against the launch's tokens a near tie may flip one and what follows
    again, _ = HFStaticDecoder(model, max_cache_len=64).generate(ids, 10, "eager")
    if int(again[1]) == int(got[1]):                   # (token 1 still came from the launch)
        assert torch.equal(again, got), (again, got)
    assert got[:2].tolist() == expected[:2].tolist()
