"""Fixed cost per tile of the tower's GEMM shapes: time(K) = overhead + slope * K for the same M x N and epilogue, on the 160x128
kernel (tile 0 picks it at M = 8224) and the 256x256 kernel (tile 512), with and without the activation.  GPU box only."""
import os, sys, json, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops

def t(M, N, K, tile, act, dt, bias=True, resid=False, iters=20, rounds=5):
    g = torch.Generator().manual_seed(1)
    a = (torch.rand(M, K, generator=g) * 2 - 1).cuda().to(dt)
    w = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).cuda().to(dt)
    b = torch.randn(N, generator=g).cuda() if bias else None
    if resid:
        out = torch.randn(M, N, generator=g).cuda()
        kw = dict(residual=out, pair_ws=ops.pair_scratch())
    else:
        out = torch.empty(M, N, device="cuda", dtype=dt); kw = {}
    for _ in range(3): ops.gemm(a, w, bias=b, act=act, out=out, tile=tile, **kw)
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): ops.gemm(a, w, bias=b, act=act, out=out, tile=tile, **kw)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return round(statistics.median(ts), 1)

if len(sys.argv) > 1 and sys.argv[1] == "tower":      # the four GEMMs of a tower layer in fp16, every kernel choice
    dt = torch.float16
    for (M, N, K, name, act, resid) in ((8224, 3072, 1024, "qkv", "none", False), (8224, 4096, 1024, "fc1", "gelu", False),
                                        (8224, 1024, 4096, "fc2", "none", True), (8224, 1024, 1024, "out", "none", True)):
        row = {"shape": name, "M": M, "N": N, "K": K}
        for tile in (0, 160, 512, 384):
            try:
                row[f"tile{tile}_us"] = t(M, N, K, tile, act, dt, resid=resid)
            except Exception as e:
                row[f"tile{tile}_us"] = str(e)[:60]
        print(json.dumps(row), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "f16c":       # the decoder's f16c GEMMs: fixed cost per launch by epilogue
    from kosmosx.model import _operand_f16c
    def tf(M, N, K, iters=10, rounds=5, **kw):
        g = torch.Generator().manual_seed(1)
        a = ops.pack_f16c_rows((torch.rand(M, K, generator=g) * 2 - 1).cuda())
        w = _operand_f16c(((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).cuda())
        extra = {}
        if kw.pop("bias", False): extra["bias"] = torch.randn(N, generator=g).cuda()
        if kw.pop("stats", False): extra["stats_out"] = torch.empty(M, N // 64, 2, device="cuda")
        if kw.pop("resid", False):
            extra.update(residual=torch.randn(M, N, generator=g).cuda(), row_stats=torch.rand(M, 2, generator=g).cuda(),
                         colsum=torch.randn(N, generator=g).cuda(), pair_ws=ops.pair_scratch())
        if kw.pop("xpos", False):
            T = 114
            extra.update(qscale=0.125, qcols=N // 3, xpos=tuple(torch.rand(T, 32, generator=g).cuda() for _ in range(4)), xpos_dim=N // 3)
        f = lambda: ops.gemm_f16c(a, w, N, K, **extra, **kw)
        for _ in range(3): f()
        ts = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): f()
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / iters * 1e3)
        return round(statistics.median(ts), 1)
    cases = [("fc1 gelu+stats -> f16c rows (EPI 7)", 3648, 8192, dict(bias=True, act="gelu", out_f16c=True, stats=True)),
             ("fc1 plain -> f16c rows (EPI 6)", 3648, 8192, dict(out_f16c=True)),
             ("fc1 bias+stats, no activation -> f16c rows (EPI 7)", 3648, 8192, dict(bias=True, out_f16c=True, stats=True)),
             ("fc1-shaped plain -> fp32 (generic)", 3648, 8192, dict()),
             ("qkv bias+qscale+xpos -> fp32 (EPI 8, 192 rows)", 3648, 6144, dict(bias=True, xpos=True)),
             ("fc2-shaped fold+bias+residual (pair split)", 3648, 2048, dict(bias=True, resid=True))]
    for name, M, N, kw in cases:
        row = {"case": name, "M": M, "N": N}
        for K in (512, 1024, 2048, 4096):
            row[f"K{K}_us"] = tf(M, N, K, **dict(kw))
        row["us_per_1024K"] = round((row["K4096_us"] - row["K2048_us"]) / 2, 1)
        row["fixed_us"] = round(row["K2048_us"] - 2 * row["us_per_1024K"], 1)
        print(json.dumps(row), flush=True)
    sys.exit(0)
for dt, dn in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
    for (M, N, name) in ((8224, 4096, "fc1"), (8224, 3072, "qkv"), (8192, 4096, "fc1p"), (8192, 3072, "qkvp")):
        for tile in (0, 512):
            for act in ("gelu", "none"):
                if act == "gelu" and "qkv" in name: continue
                row = {"dt": dn, "shape": name, "M": M, "N": N, "tile": tile, "act": act}
                for K in (512, 1024, 2048, 4096):
                    row[f"K{K}_us"] = t(M, N, K, tile, act, dt)
                row["us_per_1024K"] = round((row["K4096_us"] - row["K2048_us"]) / 2, 1)
                row["fixed_us"] = round(row["K1024_us"] - row["us_per_1024K"], 1)
                print(json.dumps(row), flush=True)
    for (M, N, K, name) in ((8224, 1024, 4096, "fc2"), (8224, 1024, 1024, "out"), (8192, 1024, 4096, "fc2p"), (8192, 1024, 1024, "outp")):
        for tile in (0, 1024) if M == 8192 else (0,):
            print(json.dumps({"dt": dn, "shape": name, "tile": tile, "resid_us": t(M, N, K, tile, "none", dt, resid=True)}), flush=True)
