"""Per-tile fixed cost of the 256x256 GEMM kernel by epilogue type (GPU box only): the same M x N problem timed at several
K; the intercept of time-per-round against K is prologue + epilogue, the slope is the main loop."""
import json, sys
from pathlib import Path
sys.path[:0] = [str(Path(__file__).resolve().parent)]
import gemm_bench as gb

M, N = 65472, 2048          # C3 rows; 256 x 8 tiles = 8 rounds of 256 workgroups
rounds = (M // 256 + (M % 256 > 0)) * (N // 256) / 256.0
for epi in ("plain", "gelu_bf16", "gelu_bf16_stats", "qkv_xpos", "resid", "resid_fold"):
    n = 6144 if epi == "qkv_xpos" else N
    rr = rounds * n / N
    pts = []
    for K in (64, 1024, 2048, 4096):
        r = gb.bench(epi, M, n, K, [512], epi=epi, iters=5, rounds=3)
        pts.append((K, r["t512_us"] / rr))
    (k0, t0), (k1, t1) = pts[1], pts[3]
    slope = (t1 - t0) / (k1 - k0)
    print(json.dumps({"epilogue": epi, "N": n, "us_per_round_by_K": {k: round(t, 1) for k, t in pts},
                      "main_loop_us_per_2048": round(slope * 2048, 1), "fixed_us_per_tile": round(t0 - slope * k0, 1)}))
