#!/bin/bash
# Round-6 measurement passes on the GPU box.  Usage: tools/r6_round.sh <stage> [<stage> ...]; outputs under gpurun_out/r6/.
# Every profile file is named for what it contains: the B = 32 kernel table profiles the step and
# nothing else (--no-extra --no-cpu-baseline --prof-steps 0), the decode tables are per precision.
#   tests      pytest -m gpu                                   -> pytest_gpu.log
#   bench      python bench.py (driver's command, 10 steps)    -> bench_default.json
#   prof32     rocprofv3 --kernel-trace --stats of the B = 32 step only -> kernel_stats_mixed_b32_step_only.csv
#   profdec    ... of tools/bench_decode.py --precision mixed / bf16    -> kernel_stats_decode_{mixed,bf16}.csv
#   profb1     ... of the batch-1 forward (bf16)                        -> kernel_stats_b1_bf16.csv
#   pmc32      tools/pmc_round.sh on the B = 32 step           -> pmc/ (summarise: tools/pmc_summary.py gpurun_out/r6/pmc profiles/r04_pmc.json)
#   pmcdecode  FETCH_SIZE / WRITE_SIZE passes of the decode step, bf16 and mixed -> pmc_decode/<mode>/
#   c3 / b1 / decode / train : the side benches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp KOSMOSX_NO_LOGGING_CONFIG=1
O=gpurun_out/r6; mkdir -p $O
stats() { # name, command...
  local name=$1; shift
  rm -rf $O/prof_$name
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$O/prof_$name" -o kx -- "$@" > "$OLDPWD/$O/prof_$name.log" 2>&1)
  local f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1)
  [[ -n "$f" ]] && cp "$f" $O/kernel_stats_$name.csv && head -8 "$f"
  rm -rf $O/prof_$name
}
for WHAT in "$@"; do
case $WHAT in
  tests)
    timeout 2400 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --durations=8 -s > $O/pytest_gpu.log 2>&1
    grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -30 ;;
  bench)
    timeout 1200 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench.err; tail -c 400 $O/bench_default.json; echo; tail -3 $O/bench.err ;;
  prof32)  stats mixed_b32_step_only python $PWD/bench.py --steps 5 --warmup 2 --pipeline 1 --objective throughput --no-cpu-baseline --no-extra --prof-steps 0 ;;   # one stream: durations are each kernel alone on the chip
  profdec) stats decode_mixed python $PWD/tools/bench_decode.py --precision mixed
           stats decode_bf16 python $PWD/tools/bench_decode.py --precision bf16 ;;
  profb1)  stats b1_bf16 python $PWD/bench.py --batch 1 --precision bf16 --pipeline 1 --steps 30 --warmup 5 --no-extra --no-cpu-baseline --prof-steps 0 ;;
  pmc32)   PMC_OUT=$O/pmc bash tools/pmc_round.sh ;;
  pmcdecode)
    for mode in bf16 mixed; do
      PMC_OUT=$O/pmc_decode/$mode PMC_GROUPS="fetch write" PMC_CMD="python $PWD/tools/bench_decode.py --precision $mode --steps 16" bash tools/pmc_round.sh
    done ;;
  c3)      timeout 400 python tools/bench_c3.py > $O/c3_bf16.json 2>/dev/null; tail -c 300 $O/c3_bf16.json; echo ;;
  b1)      timeout 300 python bench.py --batch 1 --pipeline 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b1.json 2>/dev/null; tail -c 300 $O/bench_b1.json; echo ;;
  decode)  for mode in bf16 mixed fp32; do for b in 1 4 8 16; do
             echo "== $mode B=$b"; timeout 300 python tools/bench_decode.py --precision $mode --batch $b 2>/dev/null | tail -1; done; done > $O/decode_modes.log; cat $O/decode_modes.log ;;
  train)   timeout 300 python tools/bench_train.py --precision bf16 > $O/train_bf16.json 2>/dev/null; tail -c 300 $O/train_bf16.json; echo ;;
  kloop)   timeout 900 python tools/kloop_bench.py ${KLOOP_CASES:-} > $O/kloop_bench.jsonl 2> $O/kloop_bench.err; cat $O/kloop_bench.jsonl; tail -3 $O/kloop_bench.err ;;
  testsel) # the tests named in TESTSEL (a -k expression) or the GEMM-facing files
    timeout 1500 python -m pytest ${TESTFILES:-tests/test_ops_gpu.py tests/test_f16c_gpu.py tests/test_pairk_gpu.py tests/test_xpos_kat_gpu.py} -m gpu -q --maxfail=20 -p no:cacheprovider ${TESTSEL:+-k "$TESTSEL"} > $O/pytest_sel.log 2>&1
    grep -E "passed|failed" $O/pytest_sel.log | tail -3; grep -E "^(FAILED|ERROR)" $O/pytest_sel.log | head -30 ;;
  benchq)  # headline only, both K-loop forms back to back (KOSMOSX_TUNING "14=1" = the first form)
    for arm in new old new old; do
      t=""; [[ $arm == old ]] && t="${BENCHQ_OLD:-14=1}"
      KOSMOSX_TUNING="$t" timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prof-steps 0 2>> $O/bench_quick.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'])" | tee -a $O/bench_quick.txt
    done ;;
  *) echo "unknown stage $WHAT" ;;
esac
done
echo done
