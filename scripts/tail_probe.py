import time, torch, numpy as np
n=50000
noise=torch.randn(n,3,32,32,device="cuda"); u8=torch.randint(0,256,(n,32,32,3),dtype=torch.uint8,device="cuda")
host=np.zeros((n,32,32,3),np.uint8)
def T(f,name):
    torch.cuda.synchronize(); t=time.perf_counter(); r=f(); torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter()-t)*1e3:.1f} ms", flush=True); return r
for rep in range(2):
    T(lambda: host.__setitem__(Ellipsis, u8.cpu().numpy()), "u8 cpu+assign")
    c=T(lambda: noise.cpu(), "noise.cpu() fp32 pageable")
    T(lambda: c.to(torch.float64), "cpu fp32->fp64")
    T(lambda: noise.double().cpu(), "device fp64 then cpu pageable")
    def pinned():
        r=torch.empty(n,3,32,32,dtype=torch.float64,pin_memory=True); r.copy_(noise); return r
    r=T(pinned, "pinned fp64 alloc + copy_")
    del r
    def pinned_u8():
        r=torch.empty(n,32,32,3,dtype=torch.uint8,pin_memory=True); r.copy_(u8); host[...]=r.numpy()
    T(pinned_u8, "u8 pinned + assign")
    def chunked():
        res=torch.empty(n,3,32,32,dtype=torch.float64)
        stage=[torch.empty(4096,3,32,32,dtype=torch.float64,pin_memory=True) for _ in range(2)]
        ev=[None,None]
        k=0
        pend=[]
        for lo in range(0,n,4096):
            hi=min(n,lo+4096); s=stage[k%2]
            if len(pend)>=2:
                e,plo,phi,ps=pend.pop(0); e.synchronize(); res[plo:phi].copy_(ps[:phi-plo])
            s[:hi-lo].copy_(noise[lo:hi],non_blocking=True); e=torch.cuda.Event(); e.record(); pend.append((e,lo,hi,s)); k+=1
        for e,plo,phi,ps in pend:
            e.synchronize(); res[plo:phi].copy_(ps[:phi-plo])
        return res
    T(chunked, "chunked pinned staging -> pageable fp64")
