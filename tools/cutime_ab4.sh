cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
for rnd in 1 2; do
  for arm in "|0" "|1" "15=128|0"; do
    IFS='|' read t fold <<< "$arm"
    KOSMOSX_FOLD_PRE_LN=$fold KOSMOSX_TUNING="$t" python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tuning=[$t] fold_pre_ln=$fold objective=throughput', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
