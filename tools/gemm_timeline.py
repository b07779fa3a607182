"""Where a 256x256 GEMM tile's time goes (GPU box only).  Needs the side library built with -DKX_TIMELINE
(`python kosmos-x_amd/build.py --timeline` -> kosmos-x_amd/build/tl/libkosmosx_hip_tl.so) and KOSMOSX_HIP_LIB pointing at it.  Thread 0 of every
workgroup stamps the shader clock at six points of each tile; the sums are read back per launch."""
import ctypes as C, json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("KOSMOSX_HIP_LIB", str(ROOT / "kosmos-x_amd" / "build" / "tl" / "libkosmosx_hip_tl.so"))
sys.path[:0] = [str(Path(__file__).resolve().parent)]
import torch
import gemm_bench as gb
from kosmosx import _hip

lib = C.CDLL(os.environ["KOSMOSX_HIP_LIB"])
buf = (C.c_ulonglong * 8)()
names = ["prologue (setup + first fill landed)", "K loop", "prepass (bias/act/stats on accumulators)", "park+store half 0",
         "park+store half 1"]
M = 65472
for persistent in [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0', '256'])]:
    _hip.load().kx_set_tuning(7, persistent)
    cases = (("plain", 2048, 2048), ("gelu_bf16_stats", 8192, 2048), ("qkv_xpos", 6144, 2048),
             ("resid_fold", 2048, 2048), ("resid_fold", 2048, 8192))
    if len(sys.argv) > 2:
        cases = [c for c in cases if c[0] in sys.argv[2].split(",")][:2]
    for epi, N, K in cases:
        gb.bench(epi, M, N, K, [512], epi=epi, iters=2, rounds=1)
        torch.cuda.synchronize()
        lib.kx_timeline_read(buf, 1)
        r = gb.bench(epi, M, N, K, [512], epi=epi, iters=3, rounds=1)
        torch.cuda.synchronize()
        lib.kx_timeline_read(buf, 1)
        n = max(buf[5], 1)
        rounds = (M + 255) // 256 * (N // 256) / 256.0
        print(json.dumps({"persistent": persistent, "epilogue": epi, "N": N, "K": K, "us_per_round": round(r["t512_us"] / rounds, 1),
                          "tiles_stamped": int(n), "cycles_per_tile": {names[i]: int(buf[i] / n) for i in range(5)},
                          "sum_cycles": int(sum(buf[i] for i in range(5)) / n)}))
