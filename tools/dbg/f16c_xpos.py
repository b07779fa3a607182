import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "kosmos-x_amd"); sys.path.insert(0, "tests")
from kosmosx import ops
from kosmosx.model import _operand_f16c, XPOS
g = torch.Generator().manual_seed(0)
N, K = 6144, 2048
w = (torch.randn(N, K, generator=g) * 0.03).cuda(); wp = _operand_f16c(w)
bias = torch.randn(N, generator=g).cuda()
for M in (1024, 2046):
    x = torch.randn(M, K, generator=g).cuda(); xr = ops.pack_f16c_rows(x)
    xp = XPOS(64); tabs = tuple(t.cuda() for t in (*xp.tables(M, 0, False), *xp.tables(M, 0, True)))
    print("tab max", [float(t.abs().max()) for t in tabs])
    for kw in (dict(bias=bias), dict(bias=bias, qscale=0.125, qcols=2048), dict(xpos=tabs, xpos_dim=2048)):
        for tile in (64, 128, 160, 384, 512):
            r = ops.gemm(x, w, tile=128, **kw).double(); o = ops.gemm_f16c(xr, wp, N, K, tile=tile, **kw).double()
            d = (o - r).abs()
            rows = d.amax(1); cols = d.amax(0)
            print(f"M={M} {list(kw)} tile={tile}: max err {float(d.max()):.2e} rel-to-|r| {float((d/(r.abs()+1e-3)).max()):.2e} worst row {int(rows.argmax())} col {int(cols.argmax())}; bad rows {int((rows>1e-2).sum())} bad cols {int((cols>1e-2).sum())}", flush=True)
