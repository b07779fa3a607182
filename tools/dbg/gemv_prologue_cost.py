"""How much of a tile-16 launch is its LayerNorm prologue?  Event-timed, dependent launches (same output buffer)."""
import sys
sys.path[:0] = [".", "kosmos-x_amd"]
import torch
from kosmosx import ops
dev = "cuda"
for name, N, K in (("logits", 32002, 2048), ("fc1", 8192, 2048), ("qkv", 6144, 2048)):
    ws = [(torch.randn(N, K, device=dev) / 40).bfloat16() for _ in range(6)]
    wts = [ops.tile_weight_rows(w) for w in ws]
    x = torch.randn(1, K, device=dev)
    g, b = torch.randn(K, device=dev), torch.randn(K, device=dev)
    h = ops.layernorm(x, g, b, out_dtype=torch.bfloat16)
    out = torch.empty(1, N, device=dev)
    def run(fn, n=60):
        for i in range(6): fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i % 6)
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    r = {"plain rows": run(lambda i: ops.gemm(h, ws[i], out=out, tile=16)),
         "plain tiled": run(lambda i: ops.gemm(h, wts[i], out=out, tile=16, w_tiled_rows=N)),
         "LN prologue rows": run(lambda i: ops.gemm(x, ws[i], out=out, tile=16, ln=(g, b, 1e-5))),
         "LN prologue tiled": run(lambda i: ops.gemm(x, wts[i], out=out, tile=16, ln=(g, b, 1e-5), w_tiled_rows=N))}
    print(name, {k: round(v, 1) for k, v in r.items()}, "us;  MB", round(N * K * 2 / 1e6, 1))
