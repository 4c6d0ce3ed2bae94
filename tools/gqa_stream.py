#!/usr/bin/env python3
"""The 70B launch in its measurement mode (products only, csrc/decode_block_gqa.hip dbg_layer = -2) next to the normal launch:
HIP-event time per launch and code bytes per second.  usage: python tools/gqa_stream.py [layers] [launches]
(run under rocprofv3 --pmc FETCH_SIZE for the HBM side: tools/prof_gqa_stream.sh)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quip_for_all_amd import decode as D  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 80
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
shape = D.LlamaShape(hidden=8192, ffn=28672, layers=layers, heads=64, kv_heads=8, vocab=32000)
dec = D.LlamaDecoder(shape, "E8P12", max_len=256, device="cuda:0", seed=0, device_init=True)
assert dec.block_eng and dec.eng_shape == 1
r = bench.gqa_stream_rate(dec, launches)
print(r)
