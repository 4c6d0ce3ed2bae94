"""Host-side drop-in surface: QuipQuantizer.convert_model / to_dict / from_dict / save and
load_quantized_model (reference quantizer.py:132-178, 718-756, 779-848) on a tiny HF Llama, CPU only."""
import json
import os

import pytest
import torch

transformers = pytest.importorskip("transformers")


def _tiny_config():
    from transformers import LlamaConfig
    return LlamaConfig(hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=2, vocab_size=320, max_position_embeddings=64, tie_word_embeddings=False)


def _fill_random(model, seed=0):
    from quip_for_all_amd.qlinear import QuantLinear
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, QuantLinear):
                q = m.Qidxs
                m.Qidxs.copy_(torch.randint(-32768, 32768, q.shape, generator=g, dtype=torch.int32).to(q.dtype))
                m.SU.copy_((torch.randint(0, 2, m.SU.shape, generator=g) * 2 - 1).half())
                m.SV.copy_((torch.randint(0, 2, m.SV.shape, generator=g) * 2 - 1).half())
                m.Wscale.fill_(0.02 + 0.01 * torch.rand(1, generator=g).item())
                if m.had_left is not None:
                    m.had_left.copy_(torch.linalg.qr(torch.randn(m.K_left, m.K_left, generator=g))[0].half())
                if m.had_right is not None:
                    m.had_right.copy_(torch.linalg.qr(torch.randn(m.K_right, m.K_right, generator=g))[0].half())


def test_convert_model_replaces_block_linears_only():
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer, get_layers
    from quip_for_all_amd.qlinear import QuantLinear
    model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)
    assert qz.block_name_to_quantize == "model.layers"
    ql = get_layers(model, [QuantLinear])
    assert len(ql) == 2 * 7
    assert isinstance(model.lm_head, torch.nn.Linear)
    l0 = model.model.layers[0]
    assert l0.mlp.down_proj.in_features == 688 and l0.mlp.down_proj.K_left == 43 and l0.mlp.down_proj.K_right == 1
    assert l0.self_attn.k_proj.out_features == 128
    assert qz.get_no_split_module_classes(model) == ["LlamaDecoderLayer"]


def test_config_dict_format_and_validation():
    from quip_for_all_amd.quantizer import QuipQuantizer
    d = QuipQuantizer(codebook="E8P12RVQ4B", inference=True, ft_epochs=0).to_dict()
    assert d == {"quant_method": "QUiP", "rescale_WH": False, "use_rand": True, "codebook": "E8P12RVQ4B", "codesz": 8,
                 "idx_dtype": "torch.int32", "merge_suv": False, "per_channel": False, "opt_resid_scale": -1,
                 "modules_to_not_convert": None}
    with pytest.raises(ValueError):
        QuipQuantizer(codebook="E8P13")
    with pytest.raises(ValueError):
        QuipQuantizer(codebook="E8P12", merge_suv=True, ft_epochs=3)
    with pytest.raises(NotImplementedError):
        QuipQuantizer(codebook="D4", ft_epochs=0).quantize_model(None, None)


@pytest.mark.parametrize("safe", [False, True])
def test_save_load_roundtrip(tmp_path, safe):
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer, load_quantized_model, QUIP_CONFIG, get_layers
    from quip_for_all_amd.qlinear import QuantLinear
    model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)
    _fill_random(model)
    qz.save(model, str(tmp_path), safe_serialization=safe)
    assert json.load(open(os.path.join(tmp_path, QUIP_CONFIG)))["codebook"] == "E8P12"
    with pytest.raises(RuntimeError):          # same behaviour as the reference without a GPU
        if torch.cuda.is_available():
            raise RuntimeError("gpu present")
        load_quantized_model(str(tmp_path))
    loaded = load_quantized_model(str(tmp_path), use_safetensors=safe, _require_gpu=False)
    assert loaded.is_quantized and not loaded.training
    a, b = model.state_dict(), loaded.state_dict()
    for k, v in a.items():
        if k.endswith("inv_freq"):
            continue
        assert torch.equal(v, b[k]), k
    for name, layer in get_layers(loaded, [QuantLinear]).items():
        assert abs(layer.wscale_float - float(layer.Wscale)) < 1e-9, name


def test_sharded_checkpoint_index(tmp_path):
    """the reference saves through accelerate (sharded `pytorch_model-0000x-of-0000y.bin` + index)"""
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer, load_quantized_model
    model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
    qz = QuipQuantizer(codebook="D4", inference=True, ft_epochs=0)
    qz.convert_model(model)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    keys = sorted(sd)
    half = len(keys) // 2
    shards = {"pytorch_model-00001-of-00002.bin": keys[:half], "pytorch_model-00002-of-00002.bin": keys[half:]}
    wm = {}
    for fn, ks in shards.items():
        torch.save({k: sd[k] for k in ks}, os.path.join(tmp_path, fn))
        wm.update({k: fn for k in ks})
    json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(tmp_path, "pytorch_model.bin.index.json"), "w"))
    model.config.save_pretrained(str(tmp_path))
    json.dump(qz.to_dict(), open(os.path.join(tmp_path, "quantization_config.json"), "w"))
    loaded = load_quantized_model(str(tmp_path), _require_gpu=False)
    assert loaded.model.layers[1].mlp.up_proj.Qidxs.dtype == torch.uint8
    assert torch.equal(loaded.model.layers[1].mlp.up_proj.Qidxs, sd["model.layers.1.mlp.up_proj.Qidxs"])
