import numpy as np
rng=np.random.default_rng(0)
def study(n, d, xscale, outliers=0):
    W=(rng.standard_normal((d,n))*n**-0.5).astype(np.float16).astype(np.float32)   # weights exact in fp16
    x=(rng.standard_normal(n)*xscale).astype(np.float32)
    if outliers:
        idx=rng.integers(0,n,outliers); x[idx]*=200
    ref=(W.astype(np.float64)@x.astype(np.float64))
    f32=(W@x)                                   # fp32 accumulate (what the f32 MFMA path gives, up to order)
    xh=x.astype(np.float16); xl=(x-xh.astype(np.float32)).astype(np.float16)
    hl=(W@xh.astype(np.float32))+(W@xl.astype(np.float32))
    h=(W@xh.astype(np.float32))
    bf=x.view(np.uint32); xb=((bf+0x8000)&0xffff0000).view(np.float32)             # bf16 rounding (rough)
    b=(W@xb)
    sc=np.abs(ref).max()
    return [float(np.abs(v-ref).max()/sc) for v in (f32,hl,h,b)]
print("n      d   xscale outl |  fp32      hi+lo fp16   fp16 only   bf16 only   (max |err| / max |y|)")
for n,d,xs,o in ((4096,4096,1.0,0),(4096,4096,1.0,8),(14336,4096,1.0,0),(14336,4096,0.05,0),(4096,14336,30.0,0),(4096,4096,1e-3,0),(4096,4096,300.0,4)):
    r=study(n,d,xs,o)
    print(f"{n:6d} {d:6d} {xs:7.3g} {o:4d} | "+"  ".join(f"{v:9.2e}" for v in r))
