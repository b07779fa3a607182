import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops, _hip
lib = _hip.load()
D = "cuda"
def run(a, w, t):
    q, sc, wq = ops.quantize_block16(w)
    planes = ops.tile_weight_rows_w16(q.to(D), sc.to(D))
    lib.kx_set_tuning(8, t)
    o = ops.gemm(a.to(D), planes, tile=16, w_tiled_rows=w.shape[0])
    lib.kx_set_tuning(8, 0)
    return o.cpu(), (a.double() @ wq.double().t())
torch.manual_seed(0)
M, N, K = 3, 32, 64
for name, a, w in [
    ("ones x rowconst", torch.ones(M, K), (torch.arange(N).float()[:, None] + 1).expand(N, K).contiguous() * 0.01),
    ("ones x k-ramp", torch.ones(M, K), (torch.arange(K).float()[None, :] + 1).expand(N, K).contiguous() * 0.01),
    ("rowconst x ones", (torch.arange(M).float()[:, None] + 1).expand(M, K).contiguous(), torch.ones(N, K)),
    ("k-delta a x k-ramp w", torch.eye(K)[:M] , (torch.arange(K).float()[None, :] + 1).expand(N, K).contiguous() * 0.01),
    ("k-delta5 a x k-ramp w", torch.eye(K)[5:5+M] , (torch.arange(K).float()[None, :] + 1).expand(N, K).contiguous() * 0.01),
    ("k-delta20 a x k-ramp w", torch.eye(K)[20:20+M] , (torch.arange(K).float()[None, :] + 1).expand(N, K).contiguous() * 0.01),
    ("random", torch.randn(M, K), torch.randn(N, K)),
]:
    hp, ref = run(a, w, 0)
    f32, _ = run(a, w, 5)
    print(name, "hp err", float((hp.double() - ref).abs().max()), "f32 err", float((f32.double() - ref).abs().max()))
    if float((hp.double() - ref).abs().max()) > 1e-3:
        print(" hp ", hp[:, :8]); print(" ref", ref[:, :8].float())
