#!/bin/bash
# Round-3 measurement pass: GPU parity tests, headline bench (+ rocprofv3 kernel stats of the same command), decode modes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp KOSMOSX_NO_LOGGING_CONFIG=1
TAG="${1:-a}"; O=gpurun_out/r3_$TAG; mkdir -p $O
if [[ "${2:-all}" == all || "${2:-all}" == tests ]]; then
  timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1
  grep -E "passed|failed" $O/pytest_gpu.log; grep -E "^[0-9.]+s (call|setup)" $O/pytest_gpu.log | head -8; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
fi
if [[ "${2:-all}" == all || "${2:-all}" == bench ]]; then
  timeout 1200 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench.err; tail -c 600 $O/bench_default.json; echo; tail -3 $O/bench.err
  for p in bf16 mixed fp32; do for b in 1 4 8; do timeout 300 python tools/bench_decode.py --precision $p --batch $b 2>/dev/null | tail -1; done; done > $O/decode_modes.log
  cut -c1-260 $O/decode_modes.log
  rm -rf $O/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$O/prof" -o kx -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extra --prof-steps 0 > "$OLDPWD/$O/prof_bench.log" 2>&1)
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" $O/kernel_stats_mixed_b32.csv && head -12 "$f" | cut -c1-200
  find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete 2>/dev/null
  rm -rf $O/prof_dec
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$O/prof_dec" -o kx -- python "$OLDPWD/tools/bench_decode.py" --precision mixed > "$OLDPWD/$O/prof_dec.log" 2>&1)
  f=$(find $O/prof_dec -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" $O/kernel_stats_decode_mixed.csv && head -8 "$f" | cut -c1-200
  find $O/prof_dec -name "*kernel_trace.csv" -delete; find $O/prof_dec -name "*.db" -delete 2>/dev/null
fi
echo done
if [[ "${2:-all}" == all || "${2:-all}" == pmc ]]; then
  PMC_OUT=$O/pmc bash tools/pmc_round.sh > $O/pmc_round.log 2>&1; tail -3 $O/pmc_round.log
fi
echo done-pmc
