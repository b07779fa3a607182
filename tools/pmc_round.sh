#!/bin/bash
# PMC passes for the roofline evidence (separate rocprofv3 runs per counter group; no tracing domains combined with --pmc).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp KOSMOSX_NO_LOGGING_CONFIG=1
# PMC_CMD / PMC_OUT / PMC_GROUPS select another command (e.g. the decode step), output directory and counter groups.
OUT="${PMC_OUT:-gpurun_out/pmc}"
mkdir -p $OUT
CMD="${PMC_CMD:-python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --prof-steps 0}"
GROUPS_="${PMC_GROUPS:-mfma fetch write lds}"
run() { # name, counters
  [[ " $GROUPS_ " == *" $1 "* ]] || return 0
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 -f csv -d "$OLDPWD/$OUT/$1" -o pmc -- $CMD > "$OLDPWD/$OUT/$1.log" 2>&1)
  ls $OUT/$1 2>/dev/null | head -5
}
run mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -size +30M -delete
du -sh $OUT
