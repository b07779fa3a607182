#!/bin/bash
# One GPU-box round trip: parity tests, bench, rocprofv3 kernel trace.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp KOSMOSX_NO_LOGGING_CONFIG=1
mkdir -p gpurun_out
WHAT="${1:-all}"
if [[ "$WHAT" == all || "$WHAT" == tests ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
  grep -E 'bf16:|fp32:|C1 ' gpurun_out/pytest_gpu.log | head -40; tail -25 gpurun_out/pytest_gpu.log
fi
if [[ "$WHAT" == all || "$WHAT" == bench ]]; then
  timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err
  tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
fi
if [[ "$WHAT" == all || "$WHAT" == gemm ]]; then
  timeout 600 python tools/gemm_bench.py 128,64 > gpurun_out/gemm_bench.log 2>&1
  cat gpurun_out/gemm_bench.log
fi
if [[ "$WHAT" == all || "$WHAT" == prof ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/gpurun_out/prof" -o kx -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extra --prof-steps 0 > "$OLDPWD/gpurun_out/prof_bench.log" 2>&1)
  find gpurun_out/prof -name "*stats*" | head; 
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && head -30 "$f"
  # keep the merge-back small: drop the raw trace, keep the stats
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
