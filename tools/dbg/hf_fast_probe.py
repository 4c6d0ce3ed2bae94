"""bench.hf_static_cache_extra alone (the reference's harness shape: stock HF step, captured step, the fast-decode wrapper)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from quip_for_all_amd import decode as D
print(json.dumps(bench.hf_static_cache_extra(D, "cuda:0"), indent=1))
