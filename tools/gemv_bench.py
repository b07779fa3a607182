"""A/B of the M <= 32 GEMM paths on the decode-step shapes: tile 16 (weight streaming) vs tile 64 (split-K tiles)."""
import os, sys, json, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops

SHAPES = {"qkv": (6144, 2048), "out": (2048, 2048), "fc1": (8192, 2048), "fc2": (2048, 8192), "logits": (32002, 2048)}
ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
for M in (1, 8, 16):
    for name, (N, K) in SHAPES.items():
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        # several weight copies so that consecutive launches do not find W in the 256 MB MALL / L2
        wl = [(torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(max(2, int(600e6 / (N * K * 2))))]
        out = torch.empty(M, N, device="cuda")
        r = {"M": M, "shape": name, "MB": round(N * K * 2 / 1e6, 1)}
        for tile in (16, 64):
            for w in wl[:2]:
                ops.gemm(a, w, out=out, tile=tile, **({} if tile == 16 else {'splitk_ws': ws}))
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for w in wl:
                    ops.gemm(a, w, out=out, tile=tile, **({} if tile == 16 else {'splitk_ws': ws}))
                e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) / len(wl))
            us = statistics.median(ts) * 1e3
            r[f"t{tile}_us"] = round(us, 1)
            r[f"t{tile}_TBs"] = round(N * K * 2 / us / 1e6, 2)
        print(json.dumps(r), flush=True)
