#!/bin/bash
# PMC passes for the roofline evidence (separate rocprofv3 runs per counter group; no tracing domains combined with --pmc).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp KOSMOSX_NO_LOGGING_CONFIG=1
mkdir -p gpurun_out/pmc
CMD="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --prof-steps 0"
run() { # name, counters
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 -f csv -d "$OLDPWD/gpurun_out/pmc/$1" -o pmc -- $CMD > "$OLDPWD/gpurun_out/pmc/$1.log" 2>&1)
  ls gpurun_out/pmc/$1 2>/dev/null | head -5
}
run mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"
find gpurun_out/pmc -name "*kernel_trace.csv" -delete
find gpurun_out/pmc -size +30M -delete
du -sh gpurun_out/pmc
