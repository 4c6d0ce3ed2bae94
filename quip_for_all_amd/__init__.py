"""MI355X-native QuIP# inference hot path behind the reference's operator surface
(torch.ops.quip_lib.*, qlinear.QuantLinear, quantizer.load_quantized_model)."""
from . import capi  # noqa: F401
from . import register_lib  # noqa: F401  (defines torch.ops.quip_lib.*)
from . import quant, codebook, qlinear  # noqa: F401
from .qlinear import QuantLinear  # noqa: F401
from . import quantizer  # noqa: F401
from . import quip  # noqa: F401  (layer-level quantiser: QUIP.add_batch / quant -> QuantLinear.pack)
from .quantizer import QuipQuantizer, load_quantized_model  # noqa: F401

QuipLinear = QuantLinear  # BASELINE.json's name for the same class

__all__ = ["capi", "register_lib", "quant", "codebook", "qlinear", "QuantLinear", "QuipLinear", "quantizer", "quip",
           "QuipQuantizer", "load_quantized_model"]
