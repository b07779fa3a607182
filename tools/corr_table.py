"""One fp8 correction instead of two, family by family, MEASURED on the device.

For every decoder GEMM family (qkv, out_proj, fc1, fc2, output projection) and the Perceiver, the `mixed` forward is run with
that family's KX_PREC_F16C launches contracting (a) both fp8 correction products (the shipped arithmetic), (b) the weight-side
product only, (c) the activation-side product only (kx_gemm_args.f16c_corr through tuning key 16) and compared with the fp32
CPU oracle on identical weights and inputs: max|dlogit| / rms(logits), the north star's figure of merit (bound 1e-3; the
acceptance line for shipping an assignment is 5e-4).  Workloads: C1 rows of a B = 32 multimodal batch and — with --c3 — a
KosmosLanguage row at T = 2046.  GPU box only; imports oracle/ as the checker (test infrastructure).

    python tools/corr_table.py [--c3] [--rows 0,13,31] [--out gpurun_out/corr_table.json]
"""
import argparse, json, os, sys, time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kosmos-x_amd"), str(ROOT / "tests")]
os.environ.setdefault("KOSMOSX_NO_LOGGING_CONFIG", "1")
import torch  # noqa: E402
from kosmosx import _hip  # noqa: E402
from kosmosx.model import Kosmos, KosmosLanguage  # noqa: E402
from oracle import kosmos_oracle as O  # noqa: E402
from helpers import oracle_cfg, oracle_weights, rel_err  # noqa: E402

FAMILIES = ["qkv", "out_proj", "fc1", "fc2", "logits", "perceiver"]
SIDES = {"both": 0, "weight": 1, "act": 2}


def key16(assign):
    return sum(SIDES[s] << (2 * FAMILIES.index(f)) for f, s in assign.items())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c3", action="store_true")
    ap.add_argument("--rows", default="0,13,31")
    ap.add_argument("--out", default="gpurun_out/corr_table.json")
    ap.add_argument("--combos", default="", help="extra assignments: fam=side+fam=side,fam=side ...")
    ap.add_argument("--only-base", action="store_true", help="only the 'both everywhere' row (with --tune: parity of another A/B key)")
    ap.add_argument("--tune", default="", help="kx_set_tuning pairs applied to every run, e.g. 2=6")
    a = ap.parse_args()
    lib = _hip.load()
    for kv in filter(None, a.tune.split(",")):
        lib.kx_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    rows = [int(r) for r in a.rows.split(",")]
    from kosmosx.config import DecoderConfig, KosmosConfig
    m = Kosmos._from_config(KosmosConfig(decoder=DecoderConfig()), seed=0, perturb=0.05).eval()
    g = torch.Generator().manual_seed(77)
    tok = torch.randint(0, m.cfg.vocab, (32, 50), generator=g)
    img = torch.randn(32, 3, 224, 224, generator=g)
    t0 = time.time()
    ref = O.kosmos_forward(oracle_weights(m.cpu()), tok[rows], img[rows], oracle_cfg(m.cfg), O.Switches())
    print(f"oracle C1 rows {rows}: {time.time() - t0:.0f} s", flush=True)
    m = m.to("cuda:0")
    m.precision = "mixed"
    tokd, imgd = tok.cuda(), img.cuda()

    def c1(assign):
        lib.kx_set_tuning(16, key16(assign))
        try:
            out = m(tokd, imgd)
            torch.cuda.synchronize()
            e = rel_err(out[rows], ref)
            del out
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                m(tokd, imgd)
            e1.record(); e1.synchronize()
            return e, e0.elapsed_time(e1) / 8
        finally:
            lib.kx_set_tuning(16, 0)

    cases = [{}] + ([] if a.only_base else [{f: s} for f in FAMILIES for s in ("weight", "act")])
    for combo in filter(None, a.combos.split(",")):
        cases.append(dict(kv.split("=") for kv in combo.split("+")))
    table = {"workload_c1": f"B = 32 multimodal forward, `mixed`, rows {rows} against the fp32 CPU oracle", "c1": []}
    for assign in cases:
        e, ms = c1(assign)
        table["c1"].append({"assignment": assign or "both everywhere", "max_abs_over_rms": float(f"{e:.3e}"),
                            "ms_per_step_one_stream": round(ms, 3)})
        print(f"C1  {str(assign or 'both everywhere'):42s} {e:.3e}   {ms:.2f} ms/step (one stream)", flush=True)
    del m
    torch.cuda.empty_cache()
    if a.c3:
        lm = KosmosLanguage(vocab_size=32002, dim=2048, _seed=3, _perturb=0.05).eval()
        tk = torch.randint(0, 32002, (2, 2046), generator=torch.Generator().manual_seed(5))
        t0 = time.time()
        ref3 = O.kosmos_language_forward(oracle_weights(lm.cpu()), tk[:1], O.DecoderCfg(vocab=32002))
        print(f"oracle C3 row: {time.time() - t0:.0f} s", flush=True)
        lm = lm.to("cuda:0")
        lm.precision = "f16c"
        tkd = tk.cuda()
        table["workload_c3"] = "KosmosLanguage T = 2046, `f16c`, one row (all positions) against the fp32 CPU oracle"
        table["c3"] = []
        for assign in [c for c in cases if "perceiver" not in c]:
            lib.kx_set_tuning(16, key16(assign))
            try:
                out = lm(tkd)
                torch.cuda.synchronize()
                e = rel_err(out[:1], ref3)
            finally:
                lib.kx_set_tuning(16, 0)
            table["c3"].append({"assignment": assign or "both everywhere", "max_abs_over_rms": float(f"{e:.3e}")})
            print(f"C3  {str(assign or 'both everywhere'):42s} {e:.3e}", flush=True)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
