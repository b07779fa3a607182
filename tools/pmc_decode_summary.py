"""HBM-side traffic of ONE decode step from the rocprofv3 --pmc passes of tools/bench_decode.py (tools/r4_round.sh pmcdecode):
FETCH_SIZE x2 (the guide's gfx950 correction for wide coalesced reads) + WRITE_SIZE, summed over the step's own kernels
(weight-streaming GEMVs, decode attention, the token embedding) and divided by the number of steps the run made
(= attn_decode launches / 24 layers).  Prefill / packing kernels of the same process are not counted.

    python tools/pmc_decode_summary.py gpurun_out/pmc_decode profiles/r04_decode_pmc.json  > profiles/r04_decode_pmc_summary.md
"""
import csv, json, re, sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_decode")
STEP_KERNELS = re.compile(r"(gemv_fused_kernel2?|attn_decode_kernel|embed_splice_kernel)(<[^>]*>)?")
LAYERS = 24
out = {}
print("| mode | kernel | launches / step | fetch_MB_x2 / launch | write_MB / launch |")
print("|---|---|---|---|---|")
for mode_dir in sorted(p for p in root.iterdir() if p.is_dir()):
    mode = mode_dir.name
    tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for grp in ("fetch", "write"):
        f = mode_dir / grp / "pmc_counter_collection.csv"
        if not f.exists():
            continue
        for r in csv.DictReader(open(f)):
            m = STEP_KERNELS.search(r["Kernel_Name"])
            if not m:
                continue
            k = (m.group(1) + (m.group(2) or "")).replace("unsigned short", "bf16").replace(" ", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    dec = [k for k in tot if k.startswith("attn_decode")]
    if not dec:
        continue
    steps = sum(cnt[k]["FETCH_SIZE"] for k in dec) / LAYERS
    bytes_step = 0.0
    per = {}
    for k in sorted(tot):
        n = max(cnt[k].values())
        f2, w = 2.0 * tot[k]["FETCH_SIZE"] * 1024.0, tot[k]["WRITE_SIZE"] * 1024.0
        if k.startswith("embed_splice"):            # the prefill's embedding launches belong to the prefill
            n_step = min(n, steps)
            f2, w = f2 * n_step / n, w * n_step / n
        bytes_step += (f2 + w) / steps
        per[k] = {"launches_per_step": round(n / steps, 2), "fetch_bytes_x2": round(f2 / n), "write_bytes": round(w / n)}
        print(f"| {mode} | {k} | {n / steps:.2f} | {f2 / n / 1048576:.2f} | {w / n / 1048576:.3f} |")
    out[mode] = {"bytes_per_step": round(bytes_step), "steps_in_run": steps, "kernels": per}
    print(f"| {mode} | **step total** | | **{bytes_step / 1e9:.3f} GB / step** | |")
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import gemm_sources_digest
out["_meta"] = {"gemm_sources_digest": gemm_sources_digest(),
                "how": "FETCH_SIZE x2 + WRITE_SIZE (KiB -> bytes) over the decode step's kernels / steps; tools/r4_round.sh pmcdecode"}
if len(sys.argv) > 2:
    Path(sys.argv[2]).write_text(json.dumps(out, indent=1))
