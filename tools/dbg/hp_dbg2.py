import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops, _hip
lib = _hip.load()
D = "cuda"
N, K = 32, 32
for val in (1.0, 1024.0 / 32767.0 * 3, 5.0 / 32767.0):
    w = torch.zeros(N, K); w[torch.arange(N), torch.arange(K)] = 1.0
    w2 = w.clone(); w2[:, :] = 0; w2[torch.arange(N), torch.arange(K)] = val; w2[:, (torch.arange(K) + 1) % 32] += 0  # single nonzero per row
    # keep block max = 1 so that the scale is 1/32767: put a 1.0 in a column the deltas never touch? (every k is touched) -> use val directly
    q, sc, wq = ops.quantize_block16(w2)
    planes = ops.tile_weight_rows_w16(q.to(D), sc.to(D))
    for base in (0, 16):
        a = torch.eye(K)[base:base + 16].contiguous()
        o = ops.gemm(a.to(D), planes, tile=16, w_tiled_rows=N).cpu()
        for m in range(16):
            nz = [(n, round(float(o[m, n]) / val, 3)) for n in range(N) if abs(float(o[m, n])) > 1e-12]
            print(val, "ka", base + m, "->", nz)
