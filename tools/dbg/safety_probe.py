import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quip_for_all_amd import decode as D, capi
shape = D.LlamaShape(hidden=4096, ffn=11008, layers=1, heads=32, kv_heads=32, vocab=2048)
dec = D.LlamaDecoder(shape, "E8P12", max_len=40, device="cuda:0", seed=3, device_init=True)
print("block_eng", dec.block_eng, flush=True)
dec.reset(3)
with torch.no_grad():
    l = dec.step()
torch.cuda.synchronize()
print("step ok", dec.engine_status(), flush=True)
sink = torch.zeros(4, dtype=torch.int32, device="cuda:0")
side = torch.cuda.Stream()
rc = capi.lib().quip_debug_occupy(16, 100 * 1024, ctypes.c_int64(2_000_000_000), sink.data_ptr(), side.cuda_stream)
print("occupy rc", rc, flush=True)
torch.cuda.synchronize()
print("occupy done", flush=True)
import time
t0 = time.time()
rc = capi.lib().quip_debug_occupy(16, 100 * 1024, ctypes.c_int64(int(sys.argv[1]) if len(sys.argv) > 1 else 17_000_000_000), sink.data_ptr(), side.cuda_stream)
with torch.no_grad():
    l = dec.step()
print("launched", flush=True)
try:
    torch.cuda.synchronize()
except Exception as e:
    print("sync raised", repr(e), flush=True)
print("after %.2f s: status %#x fail pos %s nan %s" % (time.time() - t0, dec.engine_status(), dec.engine_fail_position(), bool(torch.isnan(l).all())), flush=True)
dec.engine_reset()
dec.capture()
print("captured", dec.block_eng, flush=True)
rc = capi.lib().quip_debug_occupy(16, 100 * 1024, ctypes.c_int64(17_000_000_000), sink.data_ptr(), side.cuda_stream)
t0 = time.time()
for i in range(3):
    dec.graph.replay()
print("replays queued", flush=True)
torch.cuda.synchronize()
print("after %.2f s: status %#x fail pos %s" % (time.time() - t0, dec.engine_status(), dec.engine_fail_position()), flush=True)
