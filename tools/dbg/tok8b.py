#!/usr/bin/env python3
"""tokens/s of the Llama-3-8B-shaped E8P12 decoder (bench.py's llama3_8b_shape_e8p12 extra alone).  usage: tok8b.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from quip_for_all_amd import decode as D  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r = bench.time_decoder(D, D.LLAMA3_8B, "E8P12", steps, 8, "cuda:0")
print(json.dumps({k: v for k, v in r.items() if k != "gemv_roofline"}))
