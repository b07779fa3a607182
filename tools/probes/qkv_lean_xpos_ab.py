import os, sys, json, statistics
sys.path[:0] = ["/root/repo", "/root/repo/kosmos-x_amd", "/root/repo/tools"]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops, _hip
lib = _hip.load()
def run(M, N, K, T):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(bias=torch.randn(N, device="cuda"), qscale=0.125, qcols=N // 3, xpos_dim=N // 3,
              xpos=tuple(torch.rand(T, 32, device="cuda") for _ in range(4)))
    res = {}
    outs = {}
    for rnd in range(4):
        for tile in (512, 384):
            for k4 in (0, 3):
                lib.kx_set_tuning(4, k4)
                for _ in range(2): ops.gemm(a, w, out=out, tile=tile, **kw)
                outs[(tile, k4)] = out.clone()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): ops.gemm(a, w, out=out, tile=tile, **kw)
                e1.record(); e1.synchronize()
                res.setdefault((tile, k4), []).append(e0.elapsed_time(e1) / 5)
    lib.kx_set_tuning(4, 0)
    print(json.dumps({"M": M, "N": N, "K": K, **{f"t{t}_k{k}_us": round(statistics.median(v) * 1e3, 1) for (t, k), v in res.items()},
                      "lean_equals_generic_256": bool(torch.equal(outs[(512, 0)], outs[(512, 3)])),
                      "lean_equals_generic_192": bool(torch.equal(outs[(384, 0)], outs[(384, 3)]))}))
run(65472, 6144, 2048, 2046)
run(3648, 6144, 2048, 114)
