"""KX_PREC_F16C causal attention: A/B of tuning key 2 variants (0 shipped, 8 consecutive-query mapping, 6 P plain / V split,
7 P split / V plain, 4 both plain) — device time and error against a float64 softmax(QK^T)V on the same fp32 inputs.
    python tools/attn_ab.py        (GPU box only)"""
import json, os, sys, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import _hip, ops

lib = _hip.load()


def ref64(q, k, v):
    q, k, v = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))
    s = q @ k.transpose(-1, -2)
    T = s.shape[-1]
    s = s.masked_fill(torch.ones(T, T, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).flatten(2)


def timeit(fn, iters, rounds=5):
    fn(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return round(statistics.median(ts), 1)


for B, H, T, hilo in ((32, 32, 114, False), (8, 32, 2046, True), (32, 32, 2046, True)):
    g = torch.Generator().manual_seed(0)
    D = H * 64
    qkv = (torch.randn(B, T, 3 * D, generator=g) * 0.7).cuda()
    sl = lambda t, i: t[:, :, i * D:(i + 1) * D].unflatten(2, (H, 64))
    q, k, v = sl(qkv, 0) * 0.125, sl(qkv, 1), sl(qkv, 2)
    want = ref64(q[:2], k[:2], v[:2]) if T < 1000 else ref64(q[:1, :, :4], k[:1, :, :4], v[:1, :, :4])
    if hilo:       # the C3 path: q / k / v as KX_F16HL pieces (what the qkv GEMM writes at T >= 512)
        x = qkv.clone(); x[:, :, :D] *= 0.125
        s = x * 256.0
        hi = s.clamp(-65504, 65504).half(); lo = (s - hi.float()).half()
        hl = torch.cat([hi.view(B, T, 3 * H, 64), lo.view(B, T, 3 * H, 64)], -1).contiguous().view(torch.float32).view(B, T, 3 * D)
        qh, kh, vh = sl(hl, 0), sl(hl, 1), sl(hl, 2)
        f = lambda: ops.attention(qh, kh, vh, causal=True, out_f16c=True, hilo=True)
        f32 = lambda: ops.attention(qh, kh, vh, causal=True, f16c=True, hilo=True)
    else:
        f = lambda: ops.attention(q, k, v, causal=True, out_f16c=True)
        f32 = lambda: ops.attention(q, k, v, causal=True, f16c=True)
    row = {"B": B, "H": H, "T": T, "hilo": hilo}
    base = None
    for key in (0, 8, 6, 7) + (() if hilo else (4,)):
        lib.kx_set_tuning(2, key)
        try:
            us = timeit(f, 20 if T < 1000 else 3)
            out = f32()
            got = out[:2].double() if T < 1000 else out[:1].double().unflatten(2, (H, 64))[:, :, :4].flatten(2)
            err = float((got - want).abs().max() / want.pow(2).mean().sqrt())
            if key == 0: base = out.clone()
            row[f"key2={key}"] = {"us": us, "max_err_over_rms_vs_f64": float(f"{err:.2e}"), "bit_identical_to_shipped": bool(torch.equal(out, base))}
        finally:
            lib.kx_set_tuning(2, 0)
    print(json.dumps(row), flush=True)
