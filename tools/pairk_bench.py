"""Pair split of the 256x256 GEMM kernel (kx_gemm_args.pair_ws, tile 1024) against the 256x128 ring kernel (tile 256) the
decoder's N = 2048 GEMMs ran on: same operands and epilogue (folded-LN consume + bias + in-place fp32 residual), results
compared (fp32 summation order only), interleaved timing.  GPU box only.
    python tools/pairk_bench.py [f16c,f16,bf16] [shape,...]"""
import os, sys, json, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops
from kosmosx.model import _operand_f16c

SHAPES = {"dec_out_b32": (3648, 2048, 2048), "dec_fc2_b32": (3648, 2048, 8192), "dec_out_b30": (3420, 2048, 2048),
          "m3840_fc2": (3840, 2048, 8192)}

def run(kind, name, M, N, K, iters=10, rounds=5):
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(M, K, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).cuda()
    bias, colsum = torch.randn(N, generator=g).cuda(), torch.randn(N, generator=g).cuda()
    stats = torch.rand(M, 2, generator=g).cuda()
    res0 = torch.randn(M, N, generator=g).cuda()
    ws = ops.pair_scratch()
    if kind == "f16c":
        a, wp = ops.pack_f16c_rows(x), _operand_f16c(w)
        def call(tile, res):
            return ops.gemm_f16c(a, wp, N, K, bias=bias, residual=res, row_stats=stats, colsum=colsum, tile=tile, pair_ws=ws)
    else:
        dt = torch.bfloat16 if kind == "bf16" else torch.float16
        a, wd = x.to(dt), w.to(dt)
        def call(tile, res):
            return ops.gemm(a, wd, bias=bias, residual=res, out=res, row_stats=stats, colsum=colsum, tile=tile, pair_ws=ws)
    def fresh(tile):
        r = res0.clone()
        out = call(tile, r)
        return (out if kind == "f16c" else r).clone()
    ref, got = fresh(256), fresh(1024)
    torch.cuda.synchronize()
    err = float((got - ref).abs().max() / ref.abs().max())
    flags_clean = int(ws[:4096].view(torch.int32).abs().sum()) == 0
    bitrep = bool(torch.equal(fresh(1024), got))
    auto = fresh(0)
    auto_is_pair = bool(torch.equal(auto, got))
    ts = {256: [], 1024: []}
    r = res0.clone()
    for _ in range(rounds):
        for t in (256, 1024):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call(t, r)
            e1.record(); e1.synchronize()
            ts[t].append(e0.elapsed_time(e1) / iters)
    t0, t1 = statistics.median(ts[256]), statistics.median(ts[1024])
    return {"kind": kind, "shape": name, "M": M, "N": N, "K": K, "rel_err": err, "flags_clean": flags_clean,
            "bit_reproducible": bitrep, "auto_takes_pair": auto_is_pair, "ring256x128_us": round(t0 * 1e3, 1),
            "pair_us": round(t1 * 1e3, 1), "ring_tf": round(2.0 * M * N * K / t0 / 1e9, 1),
            "pair_tf": round(2.0 * M * N * K / t1 / 1e9, 1), "speedup": round(t0 / t1, 3)}

if __name__ == "__main__":
    kinds = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f16c", "f16", "bf16"]
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    for kind in kinds:
        for name, (M, N, K) in SHAPES.items():
            if only and name not in only:
                continue
            try:
                print(json.dumps(run(kind, name, M, N, K)), flush=True)
            except Exception as e:      # e.g. a shape the pair split refuses (tile 1024)
                print(json.dumps({"kind": kind, "shape": name, "error": str(e)[:200]}), flush=True)
