"""Soak of the pair split's in-launch hand-off (kx_gemm_args.pair_ws): thousands of launches over several operand sets and
both K lengths, on two streams at once (each with its own scratch) with unrelated kernels competing for the CUs, every result
compared BIT FOR BIT with the first result of its operand set (the kernel is deterministic: any stale or torn slab read
shows up as a difference) and against the unsplit kernel once.  GPU box only.   python tools/pairk_soak.py [iterations]"""
import os, sys, json, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch
from kosmosx import ops
from kosmosx.model import _operand_f16c

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g = torch.Generator().manual_seed(3)
M, N = 3648, 2048
sets = []
for K in (2048, 8192, 2048, 8192):
    x = (torch.randn(M, K, generator=g) * 1.1).cuda(); w = (torch.randn(N, K, generator=g) * 0.04).cuda()
    sets.append(dict(K=K, a=ops.pack_f16c_rows(x), wp=_operand_f16c(w), bias=torch.randn(N, generator=g).cuda(),
                     colsum=torch.randn(N, generator=g).cuda(), stats=torch.rand(M, 2, generator=g).cuda(),
                     res=torch.randn(M, N, generator=g).cuda()))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ws = [ops.pair_scratch(), ops.pair_scratch()]
noise = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
def call(s, tile, scratch):
    return ops.gemm_f16c(s["a"], s["wp"], N, s["K"], bias=s["bias"], residual=s["res"].clone(), row_stats=s["stats"],
                         colsum=s["colsum"], tile=tile, pair_ws=scratch)
first = [call(s, 1024, ws[0]) for s in sets]
ring = [call(s, 256, None) for s in sets]
torch.cuda.synchronize()
for f, r in zip(first, ring):
    assert float((f - r).abs().max()) < 2e-5 * float(r.abs().max())
bad = 0
badd = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in streams]     # every launch is compared, on its own stream, without a sync
t0 = time.time()
for it in range(iters):
    outs = []
    for si, st in enumerate(streams):
        with torch.cuda.stream(st):
            k = (it + si) % len(sets)
            if it % 3 == si:
                noise @ noise                                   # a rocBLAS GEMM of the other stream's size class competing for CUs
            outs.append((k, call(sets[k], 1024, ws[si])))
            badd[si] += (outs[-1][1] != first[k]).any()
            if it % 5 == 0:
                torch.relu_(noise)
    for k, o in outs:
        if it % 16 == 0 or it == iters - 1:                     # (comparisons synchronise: not every iteration)
            torch.cuda.synchronize()
            bad += int(not torch.equal(o, first[k]))
torch.cuda.synchronize()
bad += int(sum(int(b) for b in badd))
clean = all(int(w_[:4096].view(torch.int32).abs().sum()) == 0 for w_ in ws)
print(json.dumps({"launches": 2 * iters, "mismatches": bad, "flags_clean": clean, "seconds": round(time.time() - t0, 1)}))
assert bad == 0 and clean
