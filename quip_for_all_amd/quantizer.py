"""Inference-side surface of the reference's quantizer.py: `QuipQuantizer.convert_model`,
`to_dict` / `from_dict`, `save`, and `load_quantized_model` -- what a user needs to run an
existing QuIP-for-all checkpoint through HF transformers with the MI355X QuantLinear.

Same names, arguments, defaults, config-file format and post-load behaviour as the reference
(quantizer.py:59-178 ctor / to_dict / convert_model, :718-756 save, :779-848
load_quantized_model); the quantisation algorithm itself (`quantize_model`, quantizer.py:250-715)
is out of scope (offline, not the hot path) and raises NotImplementedError.

The checkpoint reader is self-contained (torch.load / safetensors, single file or sharded
index) so that loading does not depend on accelerate's dispatch machinery."""
import json
import os
from typing import Any, Dict, Iterable, List, Optional, Union

import torch
from torch import nn

from .codebook import codebook_id
from .qlinear import QuantLinear

QUIP_CONFIG = "quantization_config.json"      # constants.py
BLOCK_PATTERNS = ["transformer.h", "model.decoder.layers", "gpt_neox.layers", "model.layers"]   # constants.py
_CODEBOOKS = ["D4", "E8P12", "HI", "E8P12RVQ3B", "E8P12RVQ4B"]


def _conv1d_type():
    try:
        from transformers.pytorch_utils import Conv1D
        return Conv1D
    except Exception:       # transformers is only needed for the HF model classes
        return ()


def get_layers(module: nn.Module, layers: Optional[List] = None, prefix: Optional[str] = None,
               skip: Optional[List] = None, name: str = "") -> Dict[str, nn.Module]:
    """{qualified name: layer} of every layer of the given types under `prefix`, skipping names that
    contain one of `skip` (utils.py:35-74)."""
    if layers is None:
        layers = [t for t in (_conv1d_type(), nn.Conv2d, nn.Linear) if t != ()]
    skip = skip or []
    if isinstance(module, tuple(layers)):
        if (prefix is None or name.startswith(prefix)) and all(p not in name for p in skip):
            return {name: module}
        return {}
    res = {}
    for n1, child in module.named_children():
        res.update(get_layers(child, layers, prefix, skip, name + "." + n1 if name else n1))
    return res


def get_block_name_with_pattern(model: nn.Module) -> str:
    """name of the module list holding the transformer blocks (utils.py:76-93)"""
    names = [n for n, _ in model.named_modules()]
    for pat in BLOCK_PATTERNS:
        if any(pat in n for n in names):
            return pat
    raise ValueError("Block pattern could not be match. Pass `block_name_to_quantize` argument in `quantize_model`")


def recurse_getattr(obj, attr: str):
    """utils.py:141-157"""
    for a in attr.split("."):
        obj = obj[int(a)] if isinstance(obj, Iterable) and not isinstance(obj, nn.Module) or \
            isinstance(obj, (nn.ModuleList, nn.Sequential)) else getattr(obj, a)
    return obj


class QuipQuantizer:
    """Configuration holder + model converter (quantizer.py:59-178).  Accepts every constructor
    argument of the reference so that `from_dict(quantization_config)` works on real checkpoints;
    the arguments that only drive the offline quantisation are stored and otherwise unused."""

    def __init__(self, codebook: str, dataset: str = "", nsamples: int = 4096, model_seqlen: int = 2048,
                 quip_tune_iters: int = 10, sigma_reg: float = 0.01, rescale_WH: bool = False, use_rand: bool = True,
                 scale_override: float = -1, opt_resid_scale: float = -1, per_channel: bool = False,
                 block_name_to_quantize: Optional[str] = None,
                 module_name_preceding_first_block: Optional[List[str]] = None, batch_size: int = 4,
                 inference: bool = False, cache_on_gpu: bool = False, modules_to_not_convert: Optional[List] = None,
                 merge_suv: bool = False, ft_epochs: int = 5, *args, **kwargs):
        if codebook not in _CODEBOOKS:
            raise ValueError("Invalid codebook, has to be D4 or E8P12 or HI")          # quantizer.py:124-125
        if ft_epochs > 0 and merge_suv:
            raise ValueError("finetune mode is incompatible with merge_suv")           # quantizer.py:122-123
        if not (0 < sigma_reg < 1):
            raise ValueError("damp_percent must between 0 and 1.")                     # quantizer.py:129-130
        self.dataset, self.nsamples, self.model_seqlen = dataset, nsamples, model_seqlen
        self.quip_tune_iters, self.sigma_reg, self.rescale_WH = quip_tune_iters, sigma_reg, rescale_WH
        self.use_rand, self.scale_override, self.opt_resid_scale = use_rand, scale_override, opt_resid_scale
        self.per_channel, self.block_name_to_quantize = per_channel, block_name_to_quantize
        self.module_name_preceding_first_block = module_name_preceding_first_block
        self.batch_size, self.cache_on_gpu, self.merge_suv = batch_size, cache_on_gpu, merge_suv
        self.modules_to_not_convert = modules_to_not_convert
        self.ft_epochs = ft_epochs
        self.quant_method = "QUiP"
        self.codebook = codebook_id[codebook](inference=inference, opt_resid_scale=opt_resid_scale)

    def to_dict(self) -> Dict[str, Any]:
        """quantization_config.json contents (quantizer.py:132-148)"""
        return {"quant_method": "QUiP", "rescale_WH": self.rescale_WH, "use_rand": self.use_rand,
                "codebook": self.codebook.id, "codesz": self.codebook.codesz,
                "idx_dtype": str(self.codebook.idx_dtype), "merge_suv": self.merge_suv,
                "per_channel": self.per_channel, "opt_resid_scale": self.opt_resid_scale,
                "modules_to_not_convert": self.modules_to_not_convert}

    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]):
        return cls(**config_dict)                                                      # quantizer.py:150-163

    def quantize_model(self, *args, **kwargs):
        raise NotImplementedError("The model-level calibration pipeline (dataset, per-block Hessians, fine-tuning; "
                                  "quantizer.py:250-715) is outside the scope of the MI355X inference path: quantise "
                                  "with the reference and load the result with load_quantized_model().  The "
                                  "layer-level pieces are here: quip_for_all_amd.quip.QUIP (Hessian, incoherence "
                                  "processing, LDLQ on the HIP codebook search) -> QuantLinear.pack().")

    def convert_model(self, model: nn.Module) -> nn.Module:
        """replace every linear layer inside the transformer blocks by an (empty) QuantLinear
        (quantizer.py:165-178, 195-248)"""
        if self.block_name_to_quantize is None:
            self.block_name_to_quantize = get_block_name_with_pattern(model)
        names = get_layers(model, prefix=self.block_name_to_quantize, skip=self.modules_to_not_convert)
        self._replace_by_quant_layers(model, set(names))
        return model

    def get_no_split_module_classes(self, model) -> List[str]:
        return [recurse_getattr(model, self.block_name_to_quantize)[0].__class__.__name__]   # quantizer.py:180-193

    def _replace_by_quant_layers(self, module: nn.Module, names, name: str = ""):
        if isinstance(module, QuantLinear):
            return
        conv1d = _conv1d_type()
        for attr, layer in list(module.named_children()):
            full = name + "." + attr if name else attr
            if full in names:
                if isinstance(layer, nn.Linear):
                    fin, fout = layer.in_features, layer.out_features
                elif isinstance(layer, nn.Conv2d):
                    fin, fout = layer.in_channels, layer.out_channels
                elif conv1d != () and isinstance(layer, conv1d):
                    fin, fout = layer.weight.shape[0], layer.weight.shape[1]
                else:
                    continue
                device = layer.weight.device
                cb = codebook_id[self.codebook.id](inference=True, opt_resid_scale=self.opt_resid_scale)
                new = QuantLinear(fin, fout, cb, bias=(layer.bias is not None), use_rand=self.use_rand,
                                  per_channel=self.per_channel, weight_dtype=layer.weight.dtype)
                new.device = device
                if device != torch.device("meta"):
                    new = new.to(device)
                setattr(module, attr, new)
            else:
                self._replace_by_quant_layers(layer, names, full)

    def save(self, model: nn.Module, save_dir: str, max_shard_size: str = "10GB", safe_serialization: bool = False):
        """state dict + model config + quantization_config.json (quantizer.py:718-756).  Same on-disk layout as
        the reference's `Accelerator.save_model`: `pytorch_model.bin` / `model.safetensors`, or, when the state dict
        exceeds `max_shard_size`, `pytorch_model-0000i-of-0000N.bin` (`model-...safetensors`) plus the
        `*.index.json` weight map.  Tensors that share storage (tied embeddings) are written once for
        safetensors; `load_quantized_model` re-ties them."""
        os.makedirs(save_dir, exist_ok=True)
        sd, seen = {}, {}
        for k, v in model.state_dict().items():
            v = v.detach()
            if safe_serialization and v.numel() > 0:
                key = (v.device, v.data_ptr(), tuple(v.shape), tuple(v.stride()), v.dtype)
                if key in seen:
                    continue                 # tied weight: keep the first name only
                seen[key] = k
            sd[k] = v.cpu().contiguous()
        limit = _parse_size(max_shard_size)
        shards, cur, cur_bytes = [], {}, 0
        for k, v in sd.items():
            nbytes = v.numel() * v.element_size()
            if cur and cur_bytes + nbytes > limit:
                shards.append(cur)
                cur, cur_bytes = {}, 0
            cur[k] = v
            cur_bytes += nbytes
        shards.append(cur)
        base, ext = ("model", ".safetensors") if safe_serialization else ("pytorch_model", ".bin")

        def write(tensors, path):
            if safe_serialization:
                from safetensors.torch import save_file
                save_file(tensors, path, metadata={"format": "pt"})
            else:
                torch.save(tensors, path)
        if len(shards) == 1:
            write(shards[0], os.path.join(save_dir, base + ext))
        else:
            weight_map = {}
            for i, sh in enumerate(shards):
                name = f"{base}-{i + 1:05d}-of-{len(shards):05d}{ext}"
                write(sh, os.path.join(save_dir, name))
                weight_map.update({k: name for k in sh})
            total = sum(v.numel() * v.element_size() for v in sd.values())
            with open(os.path.join(save_dir, base + ext + ".index.json"), "w", encoding="utf-8") as f:
                json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2, sort_keys=True)
        if hasattr(model, "config"):
            model.config.save_pretrained(save_dir)
        with open(os.path.join(save_dir, QUIP_CONFIG), "w", encoding="utf-8") as f:
            json.dump(self.to_dict(), f, indent=2)


def _parse_size(size) -> int:
    """'10GB' / '300KB' / '5MiB' / int -> bytes (the size strings accelerate / huggingface_hub accept)"""
    if isinstance(size, int):
        return size
    s = str(size).strip().upper()
    units = [("GIB", 2 ** 30), ("MIB", 2 ** 20), ("KIB", 2 ** 10), ("GB", 10 ** 9), ("MB", 10 ** 6), ("KB", 10 ** 3),
             ("B", 1)]
    for u, m in units:
        if s.endswith(u):
            return int(float(s[:-len(u)]) * m)
    return int(s)


def _checkpoint_files(folder: str) -> List[str]:
    """weight files of a HF-style checkpoint directory (single file or sharded index)"""
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        p = os.path.join(folder, index)
        if os.path.exists(p):
            with open(p) as f:
                return sorted({os.path.join(folder, v) for v in json.load(f)["weight_map"].values()})
    for single in ("model.safetensors", "pytorch_model.bin"):
        p = os.path.join(folder, single)
        if os.path.exists(p):
            return [p]
    raise FileNotFoundError(f"no model weights (model.safetensors / pytorch_model.bin [+ index]) in {folder}")


def load_state_dict_from_folder(folder: str) -> Dict[str, torch.Tensor]:
    sd = {}
    for p in _checkpoint_files(folder):
        if p.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd.update(load_file(p))
        else:
            sd.update(torch.load(p, map_location="cpu", weights_only=True))
    return sd


def finalize_quant_layers(model: nn.Module, merge_suv: bool = False):
    """the reference's post-load step (quantizer.py:833-844): scalar Wscale folded into the input
    transform, per-channel scales normalised, all-positive SU / SV dropped for merge_suv models"""
    for layer in get_layers(model, [QuantLinear]).values():
        layer.wscale_float = layer.Wscale.mean().float().item()
        if layer.per_channel:
            layer.Wscale = layer.Wscale / layer.Wscale.mean()
        if merge_suv:
            if torch.all(layer.SU > 0):
                layer.SU = None
            if torch.all(layer.SV > 0):
                layer.SV = None


def load_quantized_model(save_folder: str, revision: Optional[str] = None,
                         torch_dtype: Optional[Union[str, torch.dtype]] = torch.float16,
                         trust_remote_code: bool = True, use_safetensors: bool = False,
                         device_map: Optional[Union[str, Dict]] = None, _require_gpu: bool = True,
                         fast_decode: Optional[bool] = None):
    """Load a QuIP-for-all checkpoint directory into a HF causal LM whose linear layers are
    QuantLinear (quantizer.py:779-848).  Like the reference: needs a GPU (raises otherwise),
    default device_map leaves the weights on the CPU ({"": "cpu"}), the model comes back in eval
    mode with `is_quantized = True`.  device_map may also be a device string or {"": device}.
    fast_decode (not in the reference; default on, QUIP_FAST_DECODE=0 or fast_decode=False switches it off): single-token
    forward calls on an initialised transformers StaticCache -- the reference's decode loop, example_generate.py:28-33, and
    `generate(cache_implementation="static")` -- run the fused decoder on the same modules and cache tensors (hf_fast.py);
    every other call is the stock forward."""
    if _require_gpu and not torch.cuda.is_available():
        raise RuntimeError("No GPU found. A GPU is needed to run quantized model.")          # quantizer.py:799-801
    if not os.path.isdir(save_folder):
        raise FileNotFoundError(f"{save_folder} is not a directory (hub download is not available: no network)")
    from transformers import AutoConfig, AutoModelForCausalLM
    config = AutoConfig.from_pretrained(save_folder, trust_remote_code=trust_remote_code, revision=revision)
    if isinstance(torch_dtype, str):
        torch_dtype = getattr(torch, torch_dtype)
    # parameters on the meta device, buffers real (rotary inv_freq, causal masks, ... are computed by the model's
    # constructor and are not in the checkpoint): what the reference's init_empty_weights(include_buffers=False)
    # does (quantizer.py:805-809)
    with _params_on_meta():
        model = AutoModelForCausalLM.from_config(config, trust_remote_code=trust_remote_code, dtype=torch_dtype)
    qcfg = getattr(config, "quantization_config", None)
    if qcfg is None:
        with open(os.path.join(save_folder, QUIP_CONFIG)) as f:
            qcfg = json.load(f)
    qcfg = dict(qcfg)
    qcfg.pop("quant_method", None)
    qcfg.pop("codesz", None)
    qcfg.pop("idx_dtype", None)
    qcfg["inference"] = True
    qcfg["ft_epochs"] = 0
    quantizer = QuipQuantizer.from_dict(qcfg)
    model = quantizer.convert_model(model)
    # materialise the meta parameters on the CPU (the QuantLinear buffers and codebook tables are real
    # already and must be kept), then fill from the checkpoint
    was_meta = _materialize_meta(model)
    sd = load_state_dict_from_folder(save_folder)
    own = model.state_dict()
    missing = [k for k in own if k not in sd]

    def soft(k):
        # the fake 0-dim `weight` of a QuantLinear; SU / SV of layers packed with merge_su / merge_sv (the reference
        # drops those parameters, qlinear.py:117-131, loads non-strictly and leaves them at their init of ones,
        # which the post-load step then removes); a tied lm_head
        # Only a merge_suv checkpoint may lack SU / SV: without it a missing SU / SV means a truncated or renamed
        # checkpoint, and filling it with ones would load a model that silently computes something else.
        stem, _, leaf = k.rpartition(".")
        if stem + ".Qidxs" in own and (leaf == "weight" or (leaf in ("SU", "SV") and bool(quantizer.merge_suv))):
            return True
        return k == "lm_head.weight" and bool(getattr(config, "tie_word_embeddings", False))
    hard_missing = [k for k in missing if not soft(k) and (k in was_meta or k.rpartition(".")[0] + ".Qidxs" in own)]
    if hard_missing:
        raise KeyError(f"checkpoint lacks {len(hard_missing)} tensors, e.g. {hard_missing[:5]}")
    cast = {}
    for k, v in sd.items():
        if k in own:
            cast[k] = v.to(own[k].dtype) if v.is_floating_point() and own[k].is_floating_point() else v
    model.load_state_dict(cast, strict=False)
    for k in missing:                      # merge_suv layers: SU / SV stay at ones
        stem, _, leaf = k.rpartition(".")
        if leaf in ("SU", "SV") and stem + ".Qidxs" in own:
            getattr(recurse_getattr(model, stem), leaf).data.fill_(1.0)
    if hasattr(model, "tie_weights") and ("lm_head.weight" in missing or getattr(config, "tie_word_embeddings", False)):
        model.tie_weights()                # materialising the meta parameters untied them
    left = [n for n, t in list(model.named_parameters()) + list(model.named_buffers()) if t.is_meta]
    if left:
        raise RuntimeError(f"tensors left on the meta device after loading: {left[:5]}")
    finalize_quant_layers(model, merge_suv=quantizer.merge_suv)
    if device_map is not None:
        dev = device_map if isinstance(device_map, (str, torch.device)) else device_map.get("", "cpu")
        if dev == "auto":
            dev = "cuda:0"
        model = model.to(dev)
    model.is_quantized = True
    model.eval()
    if fast_decode is None:
        fast_decode = os.environ.get("QUIP_FAST_DECODE", "1") != "0"
    if fast_decode:
        from .hf_fast import enable_fast_decode
        enable_fast_decode(model)
    return model


class _params_on_meta:
    """Context in which nn.Module parameters are created on the meta device while buffers stay real -- the
    behaviour of accelerate.init_empty_weights(include_buffers=False) that the reference relies on
    (quantizer.py:805), without depending on accelerate."""

    def __enter__(self):
        self._orig = nn.Module.register_parameter

        def register_parameter(module, name, param):
            self._orig(module, name, param)
            if param is not None and not param.is_meta:
                cls = type(module._parameters[name])
                kw = dict(module._parameters[name].__dict__)
                kw["requires_grad"] = param.requires_grad
                module._parameters[name] = cls(module._parameters[name].to(torch.device("meta")), **kw)
        nn.Module.register_parameter = register_parameter
        return self

    def __exit__(self, *exc):
        nn.Module.register_parameter = self._orig
        return False


def _materialize_meta(model: nn.Module, device="cpu"):
    """zero tensors for everything still on the meta device; returns the qualified names it replaced (parameters
    only after _params_on_meta; buffers too for models built under torch.device('meta'))"""
    replaced = set()
    for mname, m in model.named_modules():
        pre = mname + "." if mname else ""
        for name, p in list(m._parameters.items()):
            if p is not None and p.is_meta:
                m._parameters[name] = nn.Parameter(torch.zeros(p.shape, dtype=p.dtype, device=device),
                                                   requires_grad=p.requires_grad)
                replaced.add(pre + name)
        for name, b in list(m._buffers.items()):
            if b is not None and b.is_meta:
                m._buffers[name] = torch.zeros(b.shape, dtype=b.dtype, device=device)
                replaced.add(pre + name)
    return replaced
