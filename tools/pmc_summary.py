"""Summarises the rocprofv3 --pmc passes of tools/pmc_round.sh (gpurun_out/pmc/<group>/pmc_counter_collection.csv) into
the per-kernel table kept under profiles/ (averages per launch; units and gfx950 corrections as in
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE in KiB, FETCH_SIZE x2 for wide coalesced reads)."""
import csv, re, sys
from collections import defaultdict
from pathlib import Path
root = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")

def short(name):
    m = re.search(r"(gemm_kernel_p\d|gemm_kernel|attn_bf16_v2_kernel|attn_\w+_kernel|layernorm\w*_kernel|splitk_reduce_rows_kernel|splitk_reduce_kernel|gemv_fused_kernel2|gemv_fused_kernel|"
                  r"row_stats_finalize_kernel|embed_splice_kernel|patchify_kernel|vit_assemble_kernel)(<[^>]*>)?", name)
    if not m:
        return None
    t = m.group(2) or ""
    t = t.replace("unsigned short", "bf16").replace(" ", "")
    return m.group(1) + t

acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for grp in ("mfma", "fetch", "write", "lds"):
    f = root / grp / "pmc_counter_collection.csv"
    if not f.exists():
        continue
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
rows = []
for k in sorted(acc):
    a, c = acc[k], cnt[k]
    avg = lambda n: a[n] / c[n] if c.get(n) else 0.0
    gui = avg("GRBM_GUI_ACTIVE")
    util = 100.0 * avg("SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024) if gui else 0.0
    rows.append((k, max(c.values()), util, avg("SQ_INSTS_VALU_MFMA_MOPS_BF16"), avg("SQ_LDS_IDX_ACTIVE"), avg("SQ_LDS_BANK_CONFLICT"),
                 avg("FETCH_SIZE") / 1024, 2 * avg("FETCH_SIZE") / 1024, avg("WRITE_SIZE") / 1024))
print("| kernel | launches | MfmaUtil % | MFMA MOPS(bf16)/launch | LDS_IDX_ACTIVE | LDS_BANK_CONFLICT | fetch_MB (raw) | fetch_MB_x2 | write_MB |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {r[1]} | {r[2]:.1f} | {r[3]:.3g} | {r[4]:.3g} | {r[5]:.3g} | {r[6]:.1f} | {r[7]:.1f} | {r[8]:.1f} |")

# machine-readable companion: per-launch HBM-side traffic (fetch x2 + write, bytes) per kernel, for bench.py's roofline.traffic
import json
out = {r[0]: {"launches": r[1], "fetch_bytes_x2": round(r[7] * 1048576), "write_bytes": round(r[8] * 1048576)} for r in rows}
# which code these counters belong to: bench.py refuses the file when the GEMM-family sources have changed since
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import gemm_sources_digest
out["_meta"] = {"gemm_sources_digest": gemm_sources_digest(), "kernels": sorted(r[0] for r in rows)}
if len(sys.argv) > 2:
    Path(sys.argv[2]).write_text(json.dumps(out, indent=1))
