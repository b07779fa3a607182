"""Interleaved causal query mapping of the split-fp16 attention (tuning key 2 = 0) against 32 consecutive queries per wave (key 2 = 8),
alternating, on the C3 shape (B = 32, H = 32, T = 2046, KX_F16HL rows) and on B = 8.   python tools/attn_il_ab.py   (GPU box only)"""
import json, os, sys, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import _hip, ops
lib = _hip.load()
for B in (32, 8):
    H, T = 32, 2046
    D = H * 64
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(B, T, 3 * D, generator=g) * 0.7).cuda()
    x[:, :, :D] *= 0.125
    s = x * 256.0
    hi = s.clamp(-65504, 65504).half(); lo = (s - hi.float()).half()
    hl = torch.cat([hi.view(B, T, 3 * H, 64), lo.view(B, T, 3 * H, 64)], -1).contiguous().view(torch.float32).view(B, T, 3 * D)
    sl = lambda t, i: t[:, :, i * D:(i + 1) * D].unflatten(2, (H, 64))
    q, k, v = sl(hl, 0), sl(hl, 1), sl(hl, 2)
    f = lambda: ops.attention(q, k, v, causal=True, out_f16c=True, hilo=True)
    for _ in range(10): f()
    res = {0: [], 8: []}
    for rnd in range(5):
        for key in (0, 8):
            lib.kx_set_tuning(2, key)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); e1.synchronize()
            res[key].append(round(e0.elapsed_time(e1) * 200, 1))
    lib.kx_set_tuning(2, 0)
    print(json.dumps({"B": B, "interleaved_us": res[0], "consecutive_us": res[8]}), flush=True)
