"""kx_to_operand_pair timed alone (training step: 194 launches, ~8 ms): both outputs, straight only, transposed only."""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import grad_ops as G
dev = torch.device("cuda", 0)


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = {}
for rows, cols in [(4096, 8192), (8192, 2048), (4096, 2048), (2048, 2048), (4096, 6144)]:
    x = torch.randn(rows, cols, device=dev)
    bias = torch.empty(cols, device=dev)
    for name, kw in [("both", {}), ("straight", {"transposed": False}), ("transposed", {"straight": False}),
                     ("both+colsum", {"colsum_out": bias})]:
        us = timed(lambda: G.to_operand_pair(x, **kw))
        nout = (2 if name.startswith("both") else 1)
        out[f"{rows}x{cols}_{name}"] = {"us": round(us, 1), "TBps": round(rows * cols * (4 + 2 * nout) / us / 1e6, 2)}
print(json.dumps(out))
