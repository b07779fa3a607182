import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "kosmos-x_amd"); sys.path.insert(0, "tests")
from kosmosx import ops
def ref(q, k, v):
    q, k, v = (t.cpu().double().transpose(1, 2) for t in (q, k, v))
    s = q @ k.transpose(-1, -2)
    T = s.shape[-1]
    s = s + torch.triu(torch.full((T, T), float("-inf"), dtype=torch.float64), 1)
    return (s.softmax(-1) @ v).transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)
g = torch.Generator().manual_seed(0)
for T in (700, 1024, 1100, 1500, 2046):
    for qs, ks in ((0.3, 1.0), (0.1, 1.0), (0.3, 8.0)):
        q = (torch.randn(1, T, 2, 64, generator=g) * qs).cuda()
        k = (torch.randn(1, T, 2, 64, generator=g) * ks).cuda()
        v = (torch.randn(1, T, 2, 64, generator=g) * 2).cuda()
        r = ref(q, k, v)
        o = ops.attention(q, k, v, True, f16c=True).cpu().double()
        o32 = ops.attention(q, k, v, True).cpu().double()
        d = (o - r).abs()
        print(f"T={T} qs={qs} ks={ks}: f16c max err {float(d.max()):.3e} at row {int(d.amax(-1).argmax())}  fp32-kernel {float((o32-r).abs().max()):.3e}  nan={bool(o.isnan().any())}", flush=True)
