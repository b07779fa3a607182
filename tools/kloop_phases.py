"""Where a K-tile of the balanced 256x256 loop spends its cycles (GPU box only; side library built with
`python kosmos-x_amd/build.py --timeline`, KOSMOSX_HIP_LIB points at it).  Lane 0 of wave 0 (leading wave group) and of wave 4
(lagging group) stamp the shader clock at the start of every phase, at their arrival at the phase's closing barrier ("work")
and at its release ("span"); the table is cycles per phase, averaged over all tiles of the launch.  A phase whose work is close
to its span is the pole of that interval; the partner group's phase in the same interval is the other candidate:
    interval 1: lead R0 | lag M1      interval 2: lead M0 | lag R0      interval 3: lead R1 | lag M0      interval 4: lead M1 | lag R1
    python tools/kloop_phases.py [case,...]       (cases of tools/kloop_bench.py)"""
import ctypes as C, json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("KOSMOSX_HIP_LIB", str(ROOT / "kosmos-x_amd" / "build" / "tl" / "libkosmosx_hip_tl.so"))
sys.path[:0] = [str(Path(__file__).resolve().parent)]
import torch
import kloop_bench as kb

lib = C.CDLL(os.environ["KOSMOSX_HIP_LIB"])
KINDS = ["R0", "M0", "R1", "M1"]


def read(kind, reset):
    buf = (C.c_ulonglong * 48)()
    fn = lib.kx_timeline_phases_read_f16c if kind in ("f16c", "f16") else lib.kx_timeline_phases_read
    assert fn(buf, reset) == 0
    return list(buf)


def run(name):
    kind, epi, M, N, K, tile = kb.CASES[name]
    call = kb.make_case(kind, epi, M, N, K)
    call(tile, False); torch.cuda.synchronize(); read(kind, 1)
    tl_read = lib.kx_timeline_read if kind == "bf16" else lib.kx_timeline_read_f16c
    tl_read((C.c_ulonglong * 8)(), 1)
    if hasattr(lib, "kx_timeline_store_read_f16c"): lib.kx_timeline_store_read_f16c((C.c_ulonglong * 12)(), 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        call(tile, False)
    e1.record(); torch.cuda.synchronize()
    b = read(kind, 1)
    out = {"case": name, "us_per_call_instrumented": round(e0.elapsed_time(e1) / 3 * 1e3, 1)}
    tb = (C.c_ulonglong * 8)()
    tl_read(tb, 1)
    n = max(tb[5], 1)
    out["cycles_per_tile"] = dict(zip(("prologue", "k_loop", "prepass_or_exchange", "store_half0", "store_half1"), (int(tb[i] / n) for i in range(5))))
    out["tiles_stamped"] = int(n)
    if kind in ("f16c", "f16") and hasattr(lib, "kx_timeline_store_read_f16c"):      # three-plane tile store, per tile
        sb = (C.c_ulonglong * 12)()
        lib.kx_timeline_store_read_f16c(sb, 1)
        if sb[8]:
            names = ("h0_barrier", "h0_pack_lds_write", "h0_barrier2", "h0_row_reads_global_stores",
                     "h1_barrier", "h1_pack_lds_write", "h1_barrier2", "h1_row_reads_global_stores")
            out["f16c_store_cycles_per_tile"] = {k: int(sb[i] / sb[8]) for i, k in enumerate(names)}
    for g, gname in enumerate(("lead", "lag")):
        for seg, sname in ((0, "fp16_tiles"), (4, "fp8_tiles")):
            row = {}
            for i, ph in enumerate(KINDS):
                n = b[32 + g * 8 + seg + i]
                if n:
                    row[ph] = {"work": int(b[(g * 8 + seg + i) * 2] / n), "span": int(b[(g * 8 + seg + i) * 2 + 1] / n)}
            if row:
                row["K-tile"] = sum(v["span"] for v in row.values())
                out[f"{gname}.{sname}"] = row
    return out


if __name__ == "__main__":
    for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["dec_fc1_f16c", "c3_fc1_bf16", "sq8192_bf16", "vitp_fc1_f16"]):
        print(json.dumps(run(name)), flush=True)
