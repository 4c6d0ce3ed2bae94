"""Inference half of the reference's codebook package (codebook/__init__.py:7-13):
same class names, attributes and forward()/decompress_weight() dispatch."""
from .codebooks import (D4_codebook, E8P12_codebook, E8P12RVQ3B_codebook, E8P12RVQ4B_codebook,
                        HI4B1C_codebook)

codebook_id = {
    "D4": D4_codebook,
    "E8P12": E8P12_codebook,
    "HI": HI4B1C_codebook,
    "E8P12RVQ3B": E8P12RVQ3B_codebook,
    "E8P12RVQ4B": E8P12RVQ4B_codebook,
}
