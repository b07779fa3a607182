"""HBM-bound row kernels of the training step, timed alone: LayerNorm backward (one-pass form against the two-kernel
form, KOSMOSX_LN_BWD_TWO_KERNELS=1 selects the latter for the whole process) and the gradient-norm reduction.
    python tools/rowops_train_bench.py            # algorithmic bytes / time per launch, HIP events on the launch stream"""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import grad_ops as G

dev = torch.device("cuda", 0)
out = {"ln_bwd_form": "two kernels" if os.environ.get("KOSMOSX_LN_BWD_TWO_KERNELS") == "1" else "one pass"}


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3       # us


for rows, cols in [(4096, 2048), (4096, 8192), (32768, 2048), (8224, 1024)]:
    x, dy, dres = (torch.randn(rows, cols, device=dev) for _ in range(3))
    gam = torch.randn(cols, device=dev)
    dg, db = torch.empty(cols, device=dev), torch.empty(cols, device=dev)
    us = timed(lambda: G.layernorm_backward(x, gam, dy, dres=dres, dgamma_out=dg, dbeta_out=db))
    alg = rows * cols * 4 * 4                   # x, dy, dres read, dx written
    out[f"ln_bwd_{rows}x{cols}"] = {"us": round(us, 1), "algorithmic_MB": round(alg / 1e6, 1), "TBps": round(alg / us / 1e6, 2)}
for n in (1_270_000_000, 100_000_000):
    x = torch.randn(n, device=dev)
    us = timed(lambda: G.reduce_sum(x, squares=True), n=10, warm=2)
    out[f"sum_squares_{n}"] = {"us": round(us, 1), "TBps": round(n * 4 / us / 1e6, 2)}
for rows, cols in [(4096, 2048), (4096, 8192)]:
    xb = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
    us = timed(lambda: G.transpose(xb, 64))
    out[f"transpose_bf16_{rows}x{cols}"] = {"us": round(us, 1), "TBps": round(rows * cols * 4 / us / 1e6, 2)}
pre, dgr = torch.randn(4096, 8192, device=dev), torch.randn(4096, 8192, device=dev)
us = timed(lambda: G.gelu(pre))
out["gelu_fwd_4096x8192"] = {"us": round(us, 1), "TBps": round(pre.numel() * 8 / us / 1e6, 2)}
us = timed(lambda: G.gelu_backward(pre, dgr))
out["gelu_bwd_4096x8192"] = {"us": round(us, 1), "TBps": round(pre.numel() * 12 / us / 1e6, 2)}
print(json.dumps(out))
