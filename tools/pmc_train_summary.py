"""Per-kernel HBM traffic of the training step from the rocprofv3 --pmc passes of
    PMC_CMD="python tools/bench_train.py --precision bf16 --cpu-seconds 0 --steps 1 --warmup 1" PMC_GROUPS="fetch write lds" \
    PMC_OUT=gpurun_out/pmc_train bash tools/pmc_round.sh
(units and gfx950 corrections as tools/pmc_summary.py: FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 for wide coalesced reads)."""
import csv, re, sys
from collections import defaultdict
from pathlib import Path
root = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_train")
PAT = re.compile(r"(ln_bwd_fused_kernel|ln_bwd_fused_final_kernel|layernorm_block_kernel|layernorm_kernel|to_operand_pair_kernel|"
                 r"transpose_bf16_v8_kernel|attn_bwd_dkv_bf16_kernel|attn_bwd_dq_bf16_kernel|attn_bf16_v2_kernel|adamw_kernel|"
                 r"reduce_partial_kernel|colsum_final_kernel|cross_entropy_kernel|xpos_bwd_kernel|dropout_kernel)(<[^>]*>)?")
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for grp in ("fetch", "write", "lds"):
    f = root / grp / "pmc_counter_collection.csv"
    if not f.exists():
        continue
    for r in csv.DictReader(open(f)):
        m = PAT.search(r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1) + (m.group(2) or "").replace("unsigned short", "bf16").replace("(anonymous namespace)::", "").replace(" ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("| kernel | launches | fetch_MB (raw) | fetch_MB_x2 | write_MB | LDS_BANK_CONFLICT / LDS_IDX_ACTIVE |")
print("|---|---|---|---|---|---|")
for k in sorted(acc):
    a, c = acc[k], cnt[k]
    avg = lambda n: a[n] / c[n] if c.get(n) else 0.0
    idx = avg("SQ_LDS_IDX_ACTIVE")
    print(f"| {k} | {max(c.values())} | {avg('FETCH_SIZE') / 1024:.1f} | {2 * avg('FETCH_SIZE') / 1024:.1f} | {avg('WRITE_SIZE') / 1024:.1f} | "
          f"{(avg('SQ_LDS_BANK_CONFLICT') / idx if idx else 0):.3f} |")
