"""KX_F16HL A/B at the C3 and headline shapes: the qkv GEMM with fp32 / piece output, the split-fp16 attention on each.  GPU only."""
import os, sys, json, statistics
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops
from kosmosx.model import _operand_f16c

def timeit(fn, iters=5, rounds=3):
    fn(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return round(statistics.median(ts), 1)

for (B, T) in ((32, 2046), (32, 114)):
    H, D, K = 32, 2048, 2048
    M = B * T
    g = torch.Generator().manual_seed(0)
    a = ops.pack_f16c_rows(torch.randn(M, K, generator=g).cuda())
    w = _operand_f16c((torch.randn(3 * D, K, generator=g) * 0.02).cuda())
    bias = torch.randn(3 * D, generator=g).cuda()
    xp = tuple(torch.rand(T, 32, generator=g).cuda() + 0.5 for _ in range(4))
    kw = dict(bias=bias, qscale=0.125, qcols=D, xpos=xp, xpos_dim=D)
    r = {"B": B, "T": T}
    r["qkv_f32_us"] = timeit(lambda: ops.gemm_f16c(a, w, 3 * D, K, **kw))
    r["qkv_hl_us"] = timeit(lambda: ops.gemm_f16c(a, w, 3 * D, K, out_hilo=True, **kw))
    f32 = ops.gemm_f16c(a, w, 3 * D, K, **kw).view(B, T, 3 * D)
    hl = ops.gemm_f16c(a, w, 3 * D, K, out_hilo=True, **kw).view(B, T, 3 * D)
    sl = lambda t, i: t[:, :, i * D:(i + 1) * D].unflatten(2, (H, 64))
    r["attn_f32in_us"] = timeit(lambda: ops.attention(sl(f32, 0), sl(f32, 1), sl(f32, 2), causal=True, out_f16c=True))
    r["attn_hl_us"] = timeit(lambda: ops.attention(sl(hl, 0), sl(hl, 1), sl(hl, 2), causal=True, hilo=True, out_f16c=True))
    print(json.dumps(r), flush=True)
