import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "kosmos-x_amd"); sys.path.insert(0, "tests")
from kosmosx import ops
from kosmosx.model import _operand_f16c, XPOS
g = torch.Generator().manual_seed(0)
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().pow(2).mean().sqrt())
for M in (1024, 1500, 2046):
    for (N, K) in ((6144, 2048), (2048, 2048), (8192, 2048), (2048, 8192), (32002, 2048)):
        x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) * 0.03).cuda()
        ref = ops.gemm(x, w)
        out = ops.gemm_f16c(ops.pack_f16c_rows(x), _operand_f16c(w), N, K)
        print(f"M={M} N={N} K={K}: plain {rel(out, ref):.2e}", end="")
        if N == 6144:
            xp = XPOS(64); tabs = tuple(t.cuda() for t in (*xp.tables(M, 0, False), *xp.tables(M, 0, True)))
            bias = torch.randn(N, generator=g).cuda()
            r2 = ops.gemm(x, w, bias=bias, qscale=0.125, qcols=2048, xpos=tabs, xpos_dim=2048)
            o2 = ops.gemm_f16c(ops.pack_f16c_rows(x), _operand_f16c(w), N, K, bias=bias, qscale=0.125, qcols=2048, xpos=tabs, xpos_dim=2048)
            print(f"  xpos {rel(o2, r2):.2e} max|k| {float(r2[:, 2048:4096].abs().max()):.1f} max|q| {float(r2[:, :2048].abs().max()):.1f}", end="")
        print(flush=True)
    x = torch.randn(M, 2048, generator=g).cuda() * 2 + 0.3
    gm, bt = torch.ones(2048).cuda(), torch.zeros(2048).cuda()
    print("LN equal:", torch.equal(ops.layernorm(x, gm, bt, f16c=True), ops.pack_f16c_rows(ops.layernorm(x, gm, bt))))
