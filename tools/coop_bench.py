"""Split-K GEMM + separate reduce launch vs the in-launch reduction (kx_gemm_args.splitk_flags) on the batch-1 shapes:
device time per call (HIP events over 200 back-to-back calls), same box, alternating.   python tools/coop_bench.py   (GPU box only)"""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops
from kosmosx.model import _operand_f16c

SHAPES = [(114, 2048, 2048, "out_proj"), (114, 2048, 8192, "fc2"), (114, 8192, 2048, "fc1"), (257, 1024, 1024, "vit out_proj"),
          (257, 1024, 4096, "vit fc2"), (257, 4096, 1024, "vit fc1"), (64, 1024, 4096, "perceiver ff2")]
ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
flags = ops.splitk_flags()
for kind in ("bf16", "f16c"):
    for M, N, K, name in SHAPES:
        g = torch.Generator().manual_seed(1)
        x, w = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * 0.04).cuda()
        res = torch.randn(M, N, generator=g).cuda()
        if kind == "f16c":
            a, wp = ops.pack_f16c_rows(x), _operand_f16c(w)
            call = lambda **kw: ops.gemm_f16c(a, wp, N, K, residual=res, tile=64, splitk_ws=ws, **kw)
        else:
            a, wd = x.bfloat16(), w.bfloat16()
            call = lambda **kw: ops.gemm(a, wd, residual=res, out=res, tile=64, splitk_ws=ws, **kw)
        ts = {"two_launches": [], "in_launch": []}
        for rnd in range(3):
            for tag, kw in (("two_launches", {}), ("in_launch", {"splitk_flags": flags})):
                for _ in range(10): call(**kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200): call(**kw)
                e1.record(); e1.synchronize()
                ts[tag].append(round(e0.elapsed_time(e1) * 5, 2))      # us per call
        print(json.dumps({"kind": kind, "gemm": name, "M": M, "N": N, "K": K, **{k: min(v) for k, v in ts.items()}}), flush=True)
