"""Per-phase clocks of had_tall_batch_kernel (block 0, summed over its rows).  Needs a library whose hadamard.hip was
compiled with -DQUIP_HAD_STAMPS (QUIP_LIB_PATH=...): see tools/dbg/had_stamps.py."""
import ctypes, math, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quip_for_all_amd
from quip_for_all_amd import capi
from quip_for_all_amd.quant import get_hadK
L = capi.lib(); L_ = ctypes.CDLL(capi.LIB_PATH)
dev="cuda"; rows=32768; n=11008
op=torch.ops.quip_lib
had,K,_=get_hadK(n,True); hd=had.to(dev).half().contiguous()
x=torch.randn(rows,n,device=dev).half(); su=torch.ones(n,device=dev).half(); g=torch.randn(rows,n,device=dev).half()
def stamps():
    torch.cuda.synchronize(); out=(ctypes.c_ulonglong*16)(); assert L_.quip_had_read_stamps(out)==0
    return np.array(list(out),dtype=np.int64)[8:14]
names=["phase1","bar1","mix","bar2","fht+epi","bar3"]
for what in ("in","in+gate","out","out+res"):
    for _ in range(2):
        if what=="in": op.had_transform_fused(x,n,n,K,hd,True,su,None,None,None,0.01,None,None,1e-5,None)
        elif what=="in+gate": op.had_transform_fused(x,n,n,K,hd,True,su,None,None,None,0.01,None,None,1e-5,g)
        elif what=="out": op.had_transform_fused(x,n,n,K,hd,False,None,None,su,None,1.0,None,None,1e-5,None)
        else: op.had_transform_fused(x,n,n,K,hd,False,None,None,su,None,1.0,g,None,1e-5,None)
    s=stamps(); print(what,"total",s.sum(),dict(zip(names,(s/64).astype(int).tolist())),"per row (64 rows)")
