"""bf16 attention backward (training step) timed alone: delta + dK/dV pass + dQ pass, causal, with and without the
attention-dropout mask.  KOSMOSX_HIP_LIB points at a side library for A/B (kosmos-x_amd/build.py build_variant)."""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import grad_ops as G, ops
dev = torch.device("cuda", 0)
out = {"lib": os.environ.get("KOSMOSX_HIP_LIB", "default")}
for B, T in [(8, 512), (8, 1024), (4, 2048)]:
    Hh, D = 32, 2048
    qkv = (torch.randn(B * T, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    q3, k3, v3 = (qkv[:, i * D:(i + 1) * D].unflatten(0, (B, T)).unflatten(2, (Hh, 64)) for i in range(3))
    dout = torch.randn(B, T, D, device=dev)
    for name, drop in (("plain", None), ("dropout", (0.1, 7, 3))):
        lse = torch.empty(B, Hh, T, device=dev)
        o = ops.attention(q3, k3, v3, True, out_dtype=torch.float32, lse_out=lse, dropout=drop)
        fn = lambda: G.attention_backward(qkv, o, dout, lse, B, T, Hh, True, bf16_products=True, dropout=drop)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 7 * 2.0 * B * Hh * T * (T + 1) / 2 * 64          # S, dP (twice: both passes), dV, dK, dQ over the causal pairs
        out[f"B{B}_T{T}_{name}"] = {"us": round(us, 1), "TFLOPs": round(fl / us / 1e6, 1)}
print(json.dumps(out))
