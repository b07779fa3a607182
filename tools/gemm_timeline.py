"""Where a 256x256 GEMM tile's time goes (GPU box only).  Needs the side library built with -DKX_TIMELINE
(`python kosmos-x_amd/build.py --timeline` -> kosmos-x_amd/build/tl/libkosmosx_hip_tl.so) and KOSMOSX_HIP_LIB pointing at it.  Thread 0 of every
workgroup stamps the shader clock at six points of each tile; the sums are read back per launch."""
import ctypes as C, json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("KOSMOSX_HIP_LIB", str(ROOT / "kosmos-x_amd" / "build" / "tl" / "libkosmosx_hip_tl.so"))
sys.path[:0] = [str(Path(__file__).resolve().parent)]
import torch
import gemm_bench as gb
from kosmosx import _hip

lib = C.CDLL(os.environ["KOSMOSX_HIP_LIB"])
buf = (C.c_ulonglong * 8)()
names = ["prologue (setup + first fill landed)", "K loop", "prepass (bias/act/stats on accumulators)", "park+store half 0",
         "park+store half 1"]
if len(sys.argv) > 1 and sys.argv[1] == "pair":
    # the pair split (tile 1024) against the whole-K 256x256 walk of the same tiles (tile 512): between "K loop" and
    # "park+store half 0" sits the exchange with the partner ("prepass" column)
    import pairk_bench as pb
    from kosmosx import ops
    from kosmosx.model import _operand_f16c
    names[2] = "exchange with the partner (pair split) / prepass"
    for kind in ("f16c", "bf16"):
        for (M, N, K) in ((3648, 2048, 2048), (3648, 2048, 8192)):
            g = torch.Generator().manual_seed(1)
            x = (torch.rand(M, K, generator=g) * 2 - 1).cuda(); w = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).cuda()
            bias, colsum = torch.randn(N, generator=g).cuda(), torch.randn(N, generator=g).cuda()
            stats, res = torch.rand(M, 2, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
            ws = ops.pair_scratch()
            if kind == "f16c":
                a, wp = ops.pack_f16c_rows(x), _operand_f16c(w)
                call = lambda t: ops.gemm_f16c(a, wp, N, K, bias=bias, residual=res, row_stats=stats, colsum=colsum, tile=t, pair_ws=ws)
            else:
                a, wd = x.bfloat16(), w.bfloat16()
                call = lambda t: ops.gemm(a, wd, bias=bias, residual=res, out=res, row_stats=stats, colsum=colsum, tile=t, pair_ws=ws)
            for t in (512, 1024):
                call(t); torch.cuda.synchronize(); lib.kx_timeline_read(buf, 1)
                for _ in range(3): call(t)
                torch.cuda.synchronize(); lib.kx_timeline_read(buf, 1)
                n = max(buf[5], 1)
                print(json.dumps({"kind": kind, "M": M, "N": N, "K": K, "tile": t, "pieces_stamped": int(n),
                                  "cycles_per_piece": {names[i]: int(buf[i] / n) for i in range(5)},
                                  "sum_cycles": int(sum(buf[i] for i in range(5)) / n)}))
    sys.exit(0)
M = 65472
for persistent in [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0', '256'])]:
    _hip.load().kx_set_tuning(7, persistent)
    cases = (("plain", 2048, 2048), ("gelu_bf16_stats", 8192, 2048), ("qkv_xpos", 6144, 2048),
             ("resid_fold", 2048, 2048), ("resid_fold", 2048, 8192))
    if len(sys.argv) > 2:
        cases = [c for c in cases if c[0] in sys.argv[2].split(",")][:2]
    for epi, N, K in cases:
        gb.bench(epi, M, N, K, [512], epi=epi, iters=2, rounds=1)
        torch.cuda.synchronize()
        lib.kx_timeline_read(buf, 1)
        r = gb.bench(epi, M, N, K, [512], epi=epi, iters=3, rounds=1)
        torch.cuda.synchronize()
        lib.kx_timeline_read(buf, 1)
        n = max(buf[5], 1)
        rounds = (M + 255) // 256 * (N // 256) / 256.0
        print(json.dumps({"persistent": persistent, "epilogue": epi, "N": N, "K": K, "us_per_round": round(r["t512_us"] / rounds, 1),
                          "tiles_stamped": int(n), "cycles_per_tile": {names[i]: int(buf[i] / n) for i in range(5)},
                          "sum_cycles": int(sum(buf[i] for i in range(5)) / n)}))
