"""decode step at long contexts: 7B at 2048 / 4000, 70B at 2000 (bench.py's procedures)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from quip_for_all_amd import decode as D
print(json.dumps(bench.long_context_decode(D, "cuda:0"), indent=0))
if len(sys.argv) > 1:
    r = bench.time_decoder(D, D.LLAMA2_70B, "E8P12", 32, 8, "cuda:0")
    print({k: r[k] for k in ("tokens_per_s", "position_2000")})
