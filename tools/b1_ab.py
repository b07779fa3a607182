"""Batch-1 multimodal forward (BASELINE configs[1]): same-box alternating A/B of a tuning key, in one process.
    python tools/b1_ab.py [--key 17] [--a 0] [--b 1] [--rounds 4] [--modes bf16,mixed]       (GPU box only)"""
import argparse, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import _hip
from kosmosx.model import Kosmos

ap = argparse.ArgumentParser()
ap.add_argument("--key", type=int, default=17)
ap.add_argument("--a", type=int, default=0)
ap.add_argument("--b", type=int, default=1)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--modes", default="bf16,mixed")
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
lib = _hip.load()
m = Kosmos().eval().to("cuda:0")
g = torch.Generator().manual_seed(0)
tok = torch.randint(0, 32002, (1, 50), generator=g).cuda()
img = torch.randn(1, 3, 224, 224, generator=g).cuda()
res = {}
for mode in a.modes.split(","):
    m.precision = mode
    ts = {a.a: [], a.b: []}
    for rnd in range(a.rounds):
        for v in (a.a, a.b):
            lib.kx_set_tuning(a.key, v)
            with torch.no_grad():
                for _ in range(3):
                    m(tok, img)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    m(tok, img)
                torch.cuda.synchronize()
            ts[v].append((time.perf_counter() - t0) / a.iters * 1e3)
    lib.kx_set_tuning(a.key, 0)
    res[mode] = {f"key{a.key}={v}": [round(x, 4) for x in t] for v, t in ts.items()}
    print(mode, json.dumps(res[mode]), flush=True)
