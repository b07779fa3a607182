import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "kosmos-x_amd"); sys.path.insert(0, "tests")
from kosmosx import ops, _hip as H
from kosmosx.model import _operand_f16c, XPOS
g = torch.Generator().manual_seed(0)
N, K, M = 6144, 2048, 2046
w = (torch.randn(N, K, generator=g) * 0.03).cuda(); wp = _operand_f16c(w)
x = torch.randn(M, K, generator=g).cuda(); xr = ops.pack_f16c_rows(x)
xp = XPOS(64); tabs = tuple(t.cuda() for t in (*xp.tables(M, 0, False), *xp.tables(M, 0, True)))
kw = dict(xpos=tabs, xpos_dim=2048)
def det(f, n=8):
    outs = [f() for _ in range(n)]
    return sum(int(not torch.equal(outs[0], o)) for o in outs[1:])
xb, wb = x.bfloat16(), w.bfloat16()
for tile in (64, 128):
    print("fp32 tile", tile, "nondet:", det(lambda: ops.gemm(x, w, tile=tile, **kw)), flush=True)
    print("bf16->fp32 tile", tile, "nondet:", det(lambda: ops.gemm(xb, wb, tile=tile, out_dtype=torch.float32, **kw)), flush=True)
for tile in (64, 128, 160, 384, 512):
    print("f16c tile", tile, "nondet:", det(lambda: ops.gemm_f16c(xr, wp, N, K, tile=tile, **kw)), flush=True)
