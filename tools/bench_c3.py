"""BASELINE.json configs[2]: text-only KosmosLanguage forward, batch 32, seq 2046 (2048 overflows the reference's
2048-row position table), bf16, one MI355X.  Reports tokens/s and the fraction of the dense bf16 MFMA peak
using the causal-algorithmic flop count of SURVEY §8d (179.9 TFLOP per forward)."""
import argparse, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import _hip
from kosmosx.model import KosmosLanguage

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seq", type=int, default=2046)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--gemm-tile", type=int, default=0)
ap.add_argument("--tune", default="")
ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "fp32", "f16c", "mixed"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
m = KosmosLanguage(vocab_size=32002, dim=2048, _seed=0).eval().to(dev)     # /root/reference/example_lang.py:9-12
m.precision = a.precision
if a.gemm_tile:
    _hip.load().kx_set_tuning(1, a.gemm_tile)
for kv in filter(None, a.tune.split(",")):
    _hip.load().kx_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
tok = torch.randint(0, 32002, (a.batch, a.seq), generator=torch.Generator().manual_seed(0)).to(dev)
with torch.no_grad():
    for _ in range(a.warmup):
        out = m(tok)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = m(tok)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    _hip.prof_enable(True)
    m(tok)
    torch.cuda.synchronize()
    recs = _hip.prof_collect()
    _hip.prof_enable(False)
B, T, d, F, V, L = a.batch, a.seq, 2048, 8192, 32002, 24
flops = B * T * (L * (8 * d * d + 4 * d * F) + 2 * d * V) + B * L * 2 * d * T * (T + 1)
agg, shapes = {}, {}
for kind, x, y, z, ms in recs:
    e = agg.setdefault(kind, [0, 0.0]); e[0] += 1; e[1] += ms
    if "gemm" in str(kind):
        e = shapes.setdefault(f"{kind}:{x}x{y}x{z}", [0, 0.0, 2.0 * x * y * z]); e[0] += 1; e[1] += ms
print(json.dumps({"workload": f"KosmosLanguage forward B={B} T={T} {a.precision} (configs[2])", "ms_per_forward": round(dt * 1e3, 2),
                  "tokens_per_s": round(B * T / dt, 1), "algorithmic_tflop": round(flops / 1e12, 2),
                  "tflops": round(flops / dt / 1e12, 1), "frac_of_2.5PF": round(flops / dt / 2.5e15, 4),
                  "logits_shape": list(out.shape),
                  "kernels_ms": {k: [v[0], round(v[1], 2)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])},
                  "gemm_shapes": {k: {"n": v[0], "ms": round(v[1], 2), "tflops": round(v[2] * v[0] / v[1] / 1e9, 1)}
                                  for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])}}))
