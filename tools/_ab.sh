export KOSMOSX_NO_LOGGING_CONFIG=1
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pt.log 2>&1; grep -E "passed|failed" gpurun_out/pt.log
for v in 1 0 1 0; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --tune 4=$v > gpurun_out/bench_l$v.json 2>gpurun_out/bench_l.err; python - <<PY
import json
b=json.loads(open('gpurun_out/bench_l$v.json').read().strip().splitlines()[-1]); print('bench key4=$v', b['value'], b['ms_per_step'], [(g['M'],g['N'],g['K'],round(g['ms_per_step'],3)) for g in b['gemm_shapes'][:8]])
PY
done
for v in 1 0; do timeout 250 python tools/bench_c3.py --tune 4=$v > gpurun_out/c3_l$v.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/c3_l$v.json').read().strip().splitlines()[-1]); print('C3 key4=$v', d['ms_per_forward'], d['tflops'])
PY
done
