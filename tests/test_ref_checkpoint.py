"""On-disk round trip against checkpoints WRITTEN BY THE REFERENCE (reference
quantizer.py:718-756 save, :779-848 load_quantized_model).  tests/golden/ref_checkpoint_* were produced by the
reference's QuipQuantizer.save (accelerate's sharded .bin + index) and tests/golden/ref_checkpoint_logits.npz by the
reference's own loader + forward (tests/golden/make_ref_checkpoint.py).

  e8p12_norand     use_rand=False: K = 172 Hadamard factors are not in the checkpoint (non-persistent buffers)
  e8p12_merge_suv  merge_suv=True: no SU / SV keys in the checkpoint; K = 43 random factors stored
"""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = ("e8p12_norand", "e8p12_merge_suv")


def _dir(v):
    return os.path.join(HERE, "golden", "ref_checkpoint_" + v)


@pytest.mark.parametrize("variant", VARIANTS)
def test_reference_checkpoint_loads(variant):
    from quip_for_all_amd.qlinear import QuantLinear
    from quip_for_all_amd.quantizer import get_layers, load_quantized_model, load_state_dict_from_folder
    model = load_quantized_model(_dir(variant), _require_gpu=False)
    sd = load_state_dict_from_folder(_dir(variant))
    assert len([f for f in os.listdir(_dir(variant)) if f.endswith(".bin")]) == 4      # really sharded
    layers = get_layers(model, [QuantLinear])
    assert len(layers) == 14
    for name, layer in layers.items():
        assert torch.equal(layer.Qidxs, sd[name + ".Qidxs"])
        assert layer.wscale_float == pytest.approx(float(sd[name + ".Wscale"]))
        if variant == "e8p12_norand":
            assert (name + ".had_left") not in sd and (name + ".had_right") not in sd
            assert torch.equal(layer.SU.data, sd[name + ".SU"]) and torch.equal(layer.SV.data, sd[name + ".SV"])
            if layer.in_features == 688:
                assert layer.K_left == 172 and layer.had_left.shape == (172, 172)
                h = layer.had_left.float() * 172 ** 0.5
                assert torch.equal(h.round().abs(), torch.ones(172, 172))            # the +-1 table, scaled
            if layer.out_features == 688:
                assert layer.K_right == 172
        else:
            assert (name + ".SU") not in sd and layer.SU is None and layer.SV is None     # post-load step dropped the ones
            if layer.in_features == 688:
                assert layer.K_left == 43 and torch.equal(layer.had_left, sd[name + ".had_left"])
    # nothing left as a zero placeholder: rotary frequencies are the constructor's
    inv = model.model.rotary_emb.inv_freq
    assert float(inv[0]) == 1.0 and float(inv.min()) > 0
    emb = model.model.embed_tokens.weight
    assert torch.equal(emb, sd["model.embed_tokens.weight"]) and not bool((model.lm_head.weight == 0).all())


def test_missing_quant_tensor_is_an_error(tmp_path):
    """SU / SV may be absent (merge_suv); Qidxs may not"""
    import shutil
    from quip_for_all_amd.quantizer import load_quantized_model
    dst = tmp_path / "ck"
    shutil.copytree(_dir("e8p12_merge_suv"), dst)
    idx = json.load(open(dst / "pytorch_model.bin.index.json"))
    victim = "model.layers.1.mlp.up_proj.Qidxs"
    shard = dst / idx["weight_map"][victim]
    part = torch.load(shard, weights_only=True)
    del part[victim]
    torch.save(part, shard)
    with pytest.raises(KeyError):
        load_quantized_model(str(dst), _require_gpu=False)


@pytest.mark.parametrize("safe", [False, True])
def test_own_save_shards_and_round_trips(tmp_path, safe):
    """our writer: same file naming as accelerate's, sharded by max_shard_size, tied weights written once"""
    from quip_for_all_amd.quantizer import QuipQuantizer, load_quantized_model
    model = load_quantized_model(_dir("e8p12_merge_suv"), _require_gpu=False)
    model.config.tie_word_embeddings = True
    model.tie_weights()
    q = QuipQuantizer("E8P12", inference=True, ft_epochs=0, merge_suv=True)
    q.save(model, str(tmp_path), max_shard_size="150KB", safe_serialization=safe)
    files = sorted(os.listdir(tmp_path))
    index = "model.safetensors.index.json" if safe else "pytorch_model.bin.index.json"
    assert index in files and "quantization_config.json" in files and "config.json" in files
    shards = [f for f in files if f.endswith(".safetensors" if safe else ".bin")]
    assert len(shards) >= 3 and all("-of-%05d" % len(shards) in f for f in shards)
    wm = json.load(open(tmp_path / index))["weight_map"]
    if safe:
        assert ("lm_head.weight" in wm) != ("model.embed_tokens.weight" in wm)        # tied: stored once
    again = load_quantized_model(str(tmp_path), _require_gpu=False)
    a, b = model.state_dict(), again.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert again.lm_head.weight.data_ptr() == again.model.embed_tokens.weight.data_ptr()


# fp16 ulps of rms(logits) between our forward and the reference's own (whose CPU fp16 matmuls round differently): twice
# the maximum observed over the variants on MI355X (profiles/r03_model_ulps.txt)
_LOGIT_ULPS = 12.0      # observed: 6.0 on both reference-written checkpoints


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_reference_checkpoint_logits(variant):
    """our loader + HIP forward on the reference-written checkpoint vs the reference's loader + forward"""
    from quip_for_all_amd.quantizer import load_quantized_model
    gold = np.load(os.path.join(HERE, "golden", "ref_checkpoint_logits.npz"))
    model = load_quantized_model(_dir(variant), device_map={"": "cuda:0"})
    with torch.no_grad():
        lg = model(torch.from_numpy(gold["tokens"]).cuda()).logits.float().cpu().numpy()
    ref = gold[variant]
    assert lg.shape == ref.shape
    # both sides round activations to fp16 between layers (the reference runs fp16 matmuls on the CPU here):
    # a few fp16 ulps of the largest logit per element, and the greedy tokens agree
    scale = float(np.abs(ref).max())
    err = float(np.abs(lg - ref).max())
    rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    ulps = err / 2.0 ** (np.floor(np.log2(rms)) - 10)
    print(f"{variant}: max |logit| {scale:.3f}, rms {rms:.3f}, max abs err {err:.2e} = {ulps:.1f} fp16 ulps of rms")
    assert ulps <= _LOGIT_ULPS, ulps
    top_ref, top = ref.argmax(-1), lg.argmax(-1)
    margin = np.sort(ref, -1)[..., -1] - np.sort(ref, -1)[..., -2]
    assert np.all((top == top_ref) | (margin < 4 * err))
