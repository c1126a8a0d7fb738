import sys, torch
sys.path.insert(0, '.')
from marconet_amd import ops, packing
dev='cuda'
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
for hw,c,dt in ((128,128,torch.float16),(64,256,packing.MX_DTYPE)):
    x=torch.randn(1024,hw,hw,c,device=dev,dtype=torch.float16)
    if dt!=torch.float16: x=ops.convert(x,dt)
    wt=torch.randn(3,c,device=dev)/16; st=torch.rand(1024,c,device=dev)+0.5; bias=torch.zeros(4,device=dev)
    skip=torch.tanh(torch.randn(1024,hw//2,hw//2,4,device=dev))
    a=t(lambda: ops.torgb(x,wt,st,None,bias,skip)); b=t(lambda: ops.torgb(x,wt,st,None,bias,None))
    nb=x.numel()*x.element_size()
    print("torgb %dx%d c=%d: with skip %.3f ms (%.0f GB/s), without skip %.3f ms (%.0f GB/s)"%(hw,hw,c,a,nb/a/1e6,b,nb/b/1e6))
