"""Batch-1 forward, GEMM launches seen from the inside (needs the profiles/rejected_experiments/r02_gemm_trace.patch build): start / end of the first
and last workgroup of every tile GEMM and its split-K reduce, wall_clock64 ticks of 10 ns."""
import sys, ctypes
sys.path[:0] = [".", "kosmos-x_amd", "tests"]
import torch
from kosmosx import _hip
from kosmosx.config import DecoderConfig, KosmosConfig
from kosmosx.model import Kosmos
cfg = KosmosConfig(decoder=DecoderConfig())
m = Kosmos._from_config(cfg, seed=0).eval().cuda()
m.precision = "bf16"
g = torch.Generator().manual_seed(1)
tok = torch.randint(0, cfg.vocab, (1, 50), generator=g).cuda()
img = torch.randn(1, 3, 224, 224, generator=g).cuda()
lib = _hip.load()
buf = torch.zeros(64 * 4096, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(3): m(tok, img)
    torch.cuda.synchronize()
    lib.kx_debug_trace.argtypes = [ctypes.c_void_p]
    lib.kx_debug_trace(buf.data_ptr())
    m(tok, img)
    torch.cuda.synchronize()
    shapes = (ctypes.c_int * (4 * 4096))()
    n = lib.kx_debug_trace_shapes(shapes)
    lib.kx_debug_trace(None)
b = buf.cpu().view(-1, 64)[:n]
t_first = None
prev_end = None
tot_gemm = tot_red = tot_gap = 0
rows = []
for i in range(n):
    r = b[i]
    M, N, K = shapes[4 * i], shapes[4 * i + 1], shapes[4 * i + 2]
    g0, g1 = r[0:5].tolist(), r[8:13].tolist()
    starts = [x for x in (g0[0], g1[0]) if x]
    ends = [x for x in (g0[4], g1[4]) if x]
    if not starts: continue
    gs, ge = min(starts), max(ends)
    if t_first is None: t_first = gs
    red = None
    rs = [x for x in (r[16].item(), r[24].item()) if x]
    re_ = [x for x in (r[17].item(), r[25].item()) if x]
    if rs: red = (min(rs), max(re_), int(max(r[18].item(), r[26].item())))
    gap = (gs - prev_end) * 10 if prev_end else 0
    ph = [(x - g0[0]) * 10 if x else -1 for x in g0]
    line = f"{i:3d} {M:4d}x{N:5d}x{K:5d} start {(gs - t_first) / 100:8.2f}us gap_before {gap / 1e3:6.2f}us gemm {(ge - gs) * 10 / 1e3:6.2f}us wg0 phases {ph}"
    prev_end = ge
    tot_gemm += (ge - gs) * 10; tot_gap += gap
    if red:
        line += f" | reduce{red[2]} gap {(red[0] - ge) * 10 / 1e3:5.2f}us dur {(red[1] - red[0]) * 10 / 1e3:5.2f}us"
        tot_red += (red[1] - red[0]) * 10; tot_gap += (red[0] - ge) * 10
        prev_end = red[1]
    rows.append(line)
for l in rows[:40]: print(l)
print("...")
for l in rows[-75:-45]: print(l)
print(f"{n} GEMM launches: in GEMM kernels {tot_gemm / 1e6:.3f} ms, in reduce kernels {tot_red / 1e6:.3f} ms, between them (boundaries + other kernels) {tot_gap / 1e6:.3f} ms; span {(prev_end - t_first) / 1e5:.3f} ms")
