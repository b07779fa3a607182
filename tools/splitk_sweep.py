"""Split-K slice count of the batch-1 residual GEMMs (64 x 64 kernel + row-owning reduce with the following LayerNorm): device time
per GEMM + reduce pair for forced slice counts against the automatic choice.   python tools/splitk_sweep.py   (GPU box only)"""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops
from kosmosx.model import _operand_f16c

SHAPES = [(114, 2048, 2048, "out_proj"), (114, 2048, 8192, "fc2"), (257, 1024, 1024, "vit out_proj"), (257, 1024, 4096, "vit fc2"),
          (64, 1024, 4096, "perceiver ff2")]
ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
for kind in ("bf16", "f16c"):
    for M, N, K, name in SHAPES:
        g = torch.Generator().manual_seed(1)
        x, w = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * 0.04).cuda()
        res = torch.randn(M, N, generator=g).cuda()
        gam, bet = torch.randn(N, generator=g).cuda(), torch.randn(N, generator=g).cuda()
        if kind == "f16c":
            a, wp = ops.pack_f16c_rows(x), _operand_f16c(w)
            call = lambda sp: ops.gemm_f16c(a, wp, N, K, residual=res, tile=64, splitk_ws=ws, splitk=sp)
        else:
            a, wd = x.bfloat16(), w.bfloat16()
            call = lambda sp: ops.gemm(a, wd, residual=res, out=res, tile=64, splitk_ws=ws, splitk=sp, ln_out=(gam, bet, 1e-5, torch.bfloat16))
        row = {"kind": kind, "gemm": name, "M": M, "N": N, "K": K}
        for sp in (0, 2, 3, 4, 6, 8, 12, 16):
            try:
                ts = []
                for rnd in range(3):
                    for _ in range(10): call(sp)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(200): call(sp)
                    e1.record(); e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 5)
                row["auto" if sp == 0 else f"sp{sp}"] = round(min(ts), 2)
            except Exception as e:
                row[f"sp{sp}"] = "refused"
        print(json.dumps(row), flush=True)
