"""Which tile kernel serves a GEMM shape fastest: every `tile` value of kx_gemm on the same operands (f16c rows, fp32 output,
bias), median of interleaved rounds.  GPU box only.    python tools/tile_probe.py M,N,K [M,N,K ...]   (default: the Perceiver's shapes at B = 32)"""
import os, sys, json, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops
from kosmosx.model import _operand_f16c

TILES = (0, 64, 128, 160, 256, 384, 512, 1024)
KSPLITS = (0, 2, 4)


def probe(M, N, K, resid=False):
    g = torch.Generator().manual_seed(1)
    a = ops.pack_f16c_rows((torch.rand(M, K, generator=g) * 2 - 1).cuda())
    w = _operand_f16c(((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).cuda())
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda() if resid else None
    ws = ops.pair_scratch()
    calls = {}
    for tile in TILES:
        for ks in KSPLITS:
            if ks and tile not in (64, 128): continue
            kw = dict(bias=bias, tile=tile)
            if ks: kw["ksplit"] = ks
            if tile in (0, 1024): kw["pair_ws"] = ws
            if resid: kw["residual"] = res
            f = (lambda kw=kw: ops.gemm_f16c(a, w, N, K, **kw))
            try:
                f(); torch.cuda.synchronize()
                calls[f"tile{tile}" + (f"_ks{ks}" if ks else "")] = f
            except Exception as e:
                pass
    ts = {k: [] for k in calls}
    for _ in range(5):
        for k, f in calls.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); e1.synchronize()
            ts[k].append(e0.elapsed_time(e1) * 100)
    return {"M": M, "N": N, "K": K, "resid": resid, **{k: round(statistics.median(v), 1) for k, v in ts.items()}}


if __name__ == "__main__":
    shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]] or \
             [(2048, 1024, 4096), (10272, 1024, 1024), (2048, 4096, 1024), (2048, 512, 1024), (2048, 1024, 512), (2048, 2048, 1024)]
    for (M, N, K) in shapes:
        print(json.dumps(probe(M, N, K, resid=(N == 1024 and M == 2048))), flush=True)
