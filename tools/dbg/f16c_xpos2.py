import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "kosmos-x_amd"); sys.path.insert(0, "tests")
from kosmosx import ops, _hip as H
from kosmosx.model import _operand_f16c, XPOS
g = torch.Generator().manual_seed(0)
N = 6144
for K in (2048, 128):
    w = (torch.randn(N, K, generator=g) * 0.03).cuda(); wp = _operand_f16c(w)
    M = 2046
    x = torch.randn(M, K, generator=g).cuda(); xr = ops.pack_f16c_rows(x)
    xp = XPOS(64); tabs = tuple(t.cuda() for t in (*xp.tables(M, 0, False), *xp.tables(M, 0, True)))
    kw = dict(xpos=tabs, xpos_dim=2048)
    good = ops.gemm_f16c(xr, wp, N, K, tile=512, **kw)
    for mode in (0, 1):
        H.load().kx_set_tuning(4, mode)
        for tile in (128, 160):
            o1 = ops.gemm_f16c(xr, wp, N, K, tile=tile, **kw); o2 = ops.gemm_f16c(xr, wp, N, K, tile=tile, **kw)
            d = (o1 - good).abs()
            bad = (d > 1e-2).nonzero()
            print(f"K={K} epilogue-mode={mode} tile={tile}: deterministic={torch.equal(o1,o2)} max err {float(d.max()):.2e} nbad={len(bad)}", bad[:6].tolist(), flush=True)
            if len(bad):
                r, c = bad[0].tolist()
                print("   good", good[r, c:c+4].tolist(), "got", o1[r, c:c+4].tolist(), "plain", ops.gemm_f16c(xr, wp, N, K, tile=tile)[r, c:c+4].tolist(), "tab q/k cs", tabs[2][r, ((c & 63) >> 1)].item(), tabs[3][r, ((c&63)>>1)].item())
    H.load().kx_set_tuning(4, 0)
