"""The decoder's qkv GEMM at B = 32 (3648 x 6144 x 2048, f16c rows, bias + q-scale + XPos, fp32 out) on 192-row and 256-row tiles:
device time per launch alone on the chip.   python tools/qkv_probe.py   (GPU box only)"""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops
from kosmosx.model import _operand_f16c
M, N, K, T = 3648, 6144, 2048, 114
g = torch.Generator().manual_seed(7)
a = ops.pack_f16c_rows((torch.randn(M, K, generator=g) * 1.3).cuda())
wp = _operand_f16c((torch.randn(N, K, generator=g) * 0.05).cuda())
bias = torch.randn(N, generator=g).cuda()
tabs = tuple((torch.rand(T, 32, generator=g) * 2 - 1).cuda() for _ in range(4))
kw = dict(bias=bias, qscale=0.125, qcols=N // 3, xpos=tabs, xpos_dim=N // 3)
row = {}
for rnd in range(3):
    for tile in (384, 512):
        f = lambda: ops.gemm_f16c(a, wp, N, K, tile=tile, **kw)
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); e1.synchronize()
        row.setdefault(f"tile{tile}_us", []).append(round(e0.elapsed_time(e1) * 20, 1))
print(json.dumps(row))
