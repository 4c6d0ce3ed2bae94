import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from quip_for_all_amd import decode as D
dec = D.LlamaDecoder(D.LLAMA2_70B, "E8P12", max_len=256, device="cuda:0", seed=0, device_init=True)
print(json.dumps(bench.gqa_phase_rates(dec), indent=1))
