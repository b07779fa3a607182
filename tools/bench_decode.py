"""Incremental decoding: tokens/s of the decode step on the full-size text decoder and its
fraction of the HBM roofline (a step streams the live bf16 decoder weights once — 2.55 GB — plus the KV cache)."""
import argparse, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import _hip
from kosmosx.model import KosmosLanguage

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--prefix", type=int, default=114)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--max-len", type=int, default=512)
ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3", "f16c", "mixed"],
                help="bf16 streams bf16 weights through the tile-16 kernels; fp32 / f16c / mixed stream fp32 weights through "
                     "the same kernels on the exact-f32 MFMA (KOSMOSX_DECODE_EXACT=0: f16c keeps its tile GEMMs)")
ap.add_argument("--by-shape", action="store_true", help="also print the per-launch time of each kernel shape (in-process events)")
ap.add_argument("--tune", default="", help="A/B: kx_set_tuning key=value pairs, e.g. 1=64 (tile kernels instead of tile 16)")
a = ap.parse_args()
for kv in filter(None, a.tune.split(",")):
    _hip.load().kx_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
m = KosmosLanguage(vocab_size=32002, dim=2048, _seed=0).eval().to(dev)
m.precision = a.precision
tok = torch.randint(0, 32002, (a.batch, a.prefix + a.steps + 8), generator=torch.Generator().manual_seed(0)).to(dev)
with torch.no_grad():
    for rep in range(2):                                    # rep 0 = warm-up
        state = {"max_len": a.max_len}
        m(tok[:, : a.prefix], incremental_state=state)
        for t in range(a.prefix, a.prefix + 4):
            m(tok[:, : t + 1], incremental_state=state)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(a.prefix + 4, a.prefix + 4 + a.steps):
            out = m(tok[:, : t + 1], incremental_state=state)
        host_dt = (time.perf_counter() - t0) / a.steps      # time to ISSUE a step (host side), before the sync
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    _hip.prof_enable(True)
    m(tok[:, : a.prefix + 5 + a.steps], incremental_state=state)
    torch.cuda.synchronize()
    recs = _hip.prof_collect()
    _hip.prof_enable(False)
L, d, F, V = 24, 2048, 8192, 32002
eb = 2 if a.precision in ('bf16', 'bf16x3') else 4                 # bytes per cached key / value
ewb = eb if a.precision not in ('f16c', 'mixed') else {'1': 2.125, 'w24': 3, 'fp32': 4, '0': 4}.get(os.environ.get('KOSMOSX_DECODE_EXACT', '1'), 2.125)   # streamed per weight
wbytes = ewb * (L * (4 * d * d + 2 * d * F) + d * V)
tavg = a.prefix + 4 + a.steps / 2
kvbytes = 2 * L * a.batch * tavg * d * eb
agg = {}
shapes = {}
for kind, x, y, z, ms in recs:
    e = agg.setdefault(kind, [0, 0.0]); e[0] += 1; e[1] += ms
    e = shapes.setdefault(f"{kind} {x}x{y}x{z}", [0, 0.0]); e[0] += 1; e[1] += ms
if a.by_shape:
    for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:40s} {v[0]:3d} launches  {v[1] / v[0] * 1e3:7.2f} us each", file=sys.stderr)
print(json.dumps({"workload": f"KosmosLanguage decode step, B={a.batch}, context ~{int(tavg)} tokens, {a.precision}",
                  "ms_per_token_step": round(dt * 1e3, 3), "host_issue_ms": round(host_dt * 1e3, 3), "tokens_per_s": round(a.batch / dt, 1),
                  "bytes_per_step_GB": round((wbytes + kvbytes) / 1e9, 3),
                  "achieved_GBs": round((wbytes + kvbytes) / dt / 1e9, 1),
                  "frac_of_8TBs": round((wbytes + kvbytes) / dt / 8e12, 4),
                  "kernels_ms": {k: [v[0], round(v[1], 3)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}))
