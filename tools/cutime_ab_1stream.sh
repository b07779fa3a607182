cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
for rnd in 1 2; do
  for arm in "0 1" "15=128 1" "15=256 1" "15=384 1"; do
    set -- $arm
    t=""; [[ $1 != 0 ]] && t=$1
    KOSMOSX_TUNING="$t" python bench.py --steps 20 --warmup 5 --pipeline $2 --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tuning=${1} pipeline=${2}', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
