import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/kosmos-x_amd']
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops, _hip
def t(B,H,T,causal,n=10):
    q = (torch.randn(B, T, H, 64, device="cuda") * 0.3).to(torch.bfloat16)
    k = torch.randn(B, T, H, 64, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, T, H, 64, device="cuda").to(torch.bfloat16)
    for _ in range(3): ops.attention(q,k,v,causal)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.attention(q,k,v,causal)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for T in (512, 1024, 2046):
    print(T, 'full', round(t(8,32,T,False),1), 'causal', round(t(8,32,T,True),1))
