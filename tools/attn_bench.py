"""A/B micro-benchmark of the bf16 attention variants on the path's shapes (GPU box only)."""
import os, sys, json, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops, _hip
CASES = {"vit_b32": (32, 16, 257, 257, False), "dec_b32": (32, 32, 114, 114, True), "perc_b32": (32, 8, 64, 321, False),
         "c3_b8": (8, 32, 2046, 2046, True), "full_2046_b8": (8, 32, 2046, 2046, False)}
lib = _hip.load()
for name, (B, H, Tq, Tk, causal) in CASES.items():
    D = H * 64
    q = (torch.randn(B, Tq, H, 64, device="cuda") * 0.3).to(torch.bfloat16)
    k = torch.randn(B, Tk, H, 64, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, Tk, H, 64, device="cuda").to(torch.bfloat16)
    res = {}
    for rnd in range(3):
        for var in (0, 1):
            lib.kx_set_tuning(2, var)
            for _ in range(2): ops.attention(q, k, v, causal)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.attention(q, k, v, causal)
            e1.record(); e1.synchronize()
            res.setdefault(var, []).append(e0.elapsed_time(e1) / 10)
    lib.kx_set_tuning(2, 0)
    fl = 4.0 * B * H * Tq * Tk * 64 * (0.5 * (Tk + 1) / Tk if causal else 1.0)   # causal: algorithmic T(T+1)/2 pairs
    print(json.dumps({"case": name, "B": B, "H": H, "Tq": Tq, "Tk": Tk, "causal": causal,
                      **{f"v{2 - v}_us": round(statistics.median(t) * 1e3, 1) for v, t in res.items()},
                      **{f"v{2 - v}_tf": round(fl / statistics.median(t) / 1e9, 1) for v, t in res.items()}}), flush=True)
