"""Sustained kx_gemm loop (default 8192^3 bf16, ~6 s) for tools/power_probe.sh: what clock / power does the chip hold
under back-to-back MFMA work?  Prints the TFLOP/s of each second of the run."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 512
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
a = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16)
w = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16)
out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(a, w, out=out, tile=tile)
torch.cuda.synchronize()
t_end = time.perf_counter() + secs
while time.perf_counter() < t_end:
    t0 = time.perf_counter()
    for _ in range(200):
        ops.gemm(a, w, out=out, tile=tile)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{2.0 * n ** 3 * 200 / dt / 1e12:.1f} TFLOP/s", flush=True)
