"""A/B micro-benchmark of the LayerNorm variants on the path's shapes (GPU box only)."""
import os, sys, json, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops, _hip
SHAPES = [(8224, 1024), (3648, 2048), (3648, 8192), (257, 1024), (114, 2048), (114, 8192), (65472, 2048)]
lib = _hip.load()
for rows, cols in SHAPES:
    x = torch.randn(rows, cols, device="cuda"); g = torch.ones(cols, device="cuda"); b = torch.zeros(cols, device="cuda")
    out = torch.empty(rows, cols, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for var in (0, 1):
            lib.kx_set_tuning(0, var)
            for _ in range(3): ops.layernorm(x, g, b, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.layernorm(x, g, b, out=out)
            e1.record(); e1.synchronize()
            res.setdefault(var, []).append(e0.elapsed_time(e1) / 20)
    lib.kx_set_tuning(0, 0)
    print(json.dumps({"rows": rows, "cols": cols, **{f"v{v}_us": round(statistics.median(t) * 1e3, 2) for v, t in res.items()},
                      **{f"v{v}_gbs": round(rows * cols * 6 / statistics.median(t) / 1e6, 0) for v, t in res.items()}}), flush=True)
