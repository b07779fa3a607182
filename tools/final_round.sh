#!/bin/bash
# End-of-round measurement pass on the GPU box: parity tests, headline bench (+ rocprofv3 kernel stats of the same
# command), C3, batch-1 / decode, training step.  Outputs under gpurun_out/final/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp KOSMOSX_NO_LOGGING_CONFIG=1
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log; grep -E "^[0-9.]+s (call|setup)" $O/pytest_gpu.log | head -8
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench.err; tail -c 300 $O/bench_default.json; echo
timeout 300 python tools/bench_c3.py > $O/c3_bf16.json 2>/dev/null
timeout 300 python bench.py --batch 1 --pipeline 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b1.json 2>/dev/null
timeout 300 python tools/bench_decode.py > $O/decode.log 2>&1
timeout 300 python tools/bench_decode.py --batch 4 > $O/decode_b4.log 2>&1
timeout 300 python tools/bench_decode.py --tune 8=1 > $O/decode_first_form.log 2>&1
timeout 300 python tools/bench_train.py --precision bf16 > $O/train_bf16.json 2>/dev/null
timeout 300 python tools/bench_train.py --precision bf16 --batch 32 --seq 1024 > $O/train_bf16_b32.json 2>/dev/null
timeout 300 python tools/bench_train.py --precision bf16 --train-mode --cpu-seconds 0 > $O/train_bf16_trainmode.json 2>/dev/null
timeout 300 python tools/bench_train.py --precision bf16 --train-mode --cpu-seconds 0 --batch 32 --seq 1024 > $O/train_bf16_trainmode_b32.json 2>/dev/null
timeout 200 python tools/rowops_train_bench.py > $O/rowops_train.json 2>/dev/null
rm -rf $O/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$O/prof" -o kx -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --prof-steps 0 > "$OLDPWD/$O/prof_bench.log" 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" $O/kernel_stats_b32.csv && head -12 "$f"
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete 2>/dev/null
for what in dec b1; do
  rm -rf $O/prof_$what
  if [[ $what == dec ]]; then cmd="python $PWD/tools/bench_decode.py"; else cmd="python $PWD/bench.py --batch 1 --precision bf16 --pipeline 1 --steps 30 --warmup 5 --no-extra --no-cpu-baseline --prof-steps 0"; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OLDPWD/$O/prof_$what" -o kx -- $cmd > "$OLDPWD/$O/prof_$what.log" 2>&1)
  f=$(find $O/prof_$what -name "*kernel_stats.csv" | head -1); [[ -n "$f" ]] && cp "$f" $O/kernel_stats_$what.csv && head -6 "$f"
  find $O/prof_$what -name "*kernel_trace.csv" -delete; find $O/prof_$what -name "*.db" -delete 2>/dev/null
done
echo done
