import sys, json, torch
sys.path.insert(0, '.')
from open_flamingo_amd.hip.ops import Ops, BF16
from open_flamingo_amd.hip import abi
ops = Ops.default()
dev='cuda'
filler_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); filler_b = torch.empty_like(filler_a)
def timed(fn, cold=True, n=20):
    ts=[]
    for i in range(n):
        if cold: filler_b.copy_(filler_a)
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    ts=sorted(ts[3:]); return round(ts[len(ts)//2],1)
for M in (64, 128):
  for N,K in ((3072,1024),(1024,1024),(4096,1024),(1024,4096)):
    x=torch.randn(M,K,device=dev).to(BF16); w=torch.randn(N,K,device=dev).to(BF16); b=torch.randn(N,device=dev).to(BF16)
    bias=b.expand(M,N).contiguous(); out=torch.empty(M,N,device=dev,dtype=BF16)
    f_ours=lambda: ops.gemm(x,w,out,epi=abi.EPI_GATE_RESID,aux=bias)
    f_plain=lambda: ops.gemm(x,w,out)
    f_vendor=lambda: torch.addmm(b,x,w.t())
    f_ours(); ref=torch.addmm(b,x,w.t()); 
    err=(out.float()-ref.float()).abs().max().item()/ref.float().abs().max().item()
    print(json.dumps({"M":M,"N":N,"K":K,"ours_bias_us":timed(f_ours),"ours_plain_us":timed(f_plain),"vendor_addmm_us":timed(f_vendor),"ours_bias_hot":timed(f_ours,False),"vendor_hot":timed(f_vendor,False),"rel_err":err}),flush=True)
