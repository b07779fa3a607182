import sys
sys.path[:0] = [".", "kosmos-x_amd"]
import torch
from kosmosx import ops
torch.manual_seed(0)
for M, N, K in ((114, 2048, 8192), (114, 8192, 2048), (114, 6144, 2048), (3648, 2048, 8192), (3648, 8192, 2048), (257, 1024, 4096)):
    a = torch.randn(M, K) + 0.3          # activations with a mean (post-GELU like)
    w = torch.randn(N, K) / K ** 0.5
    ref = a.double() @ w.double().t()
    rms = ref.pow(2).mean().sqrt()
    cpu = a @ w.t()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    hip = ops.gemm(a.cuda(), w.cuda(), splitk_ws=ws).cpu()
    hip128 = ops.gemm(a.cuda(), w.cuda(), tile=128).cpu()
    f = lambda x: float((x.double() - ref).pow(2).mean().sqrt() / rms)
    print(f"M={M} N={N} K={K}: rms err vs f64  cpu {f(cpu):.2e}  hip(auto) {f(hip):.2e}  hip(tile128, one K chain) {f(hip128):.2e}")
