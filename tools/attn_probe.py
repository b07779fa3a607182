import os, sys, json, statistics
sys.path[:0] = ["/root/repo", "/root/repo/kosmos-x_amd"]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops
def timeit(fn, iters=20, rounds=5):
    fn(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return round(statistics.median(ts), 1)
for T, H, causal, kind in ((114, 32, True, "f16c"), (257, 16, False, "f16"), (114, 32, True, "f16"), (64, 32, True, "f16c"), (128, 32, True, "f16c")):
    row = {"T": T, "H": H, "kind": kind}
    for B in (4, 8, 16, 32, 64):
        g = torch.Generator().manual_seed(0)
        qkv = torch.randn(B, T, 3 * H * 64, generator=g).cuda()
        if kind == "f16": qkv = qkv.half()
        D = H * 64
        sl = lambda i: qkv[:, :, i * D:(i + 1) * D].unflatten(2, (H, 64))
        if kind == "f16c":
            f = lambda: ops.attention(sl(0), sl(1), sl(2), causal=causal, out_f16c=True)
        else:
            f = lambda: ops.attention(sl(0), sl(1), sl(2), causal=causal)
        row[f"B{B}_us"] = timeit(f)
    print(json.dumps(row), flush=True)
