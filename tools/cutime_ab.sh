#!/bin/bash
# "CU-time" experiment (round 6): with two or three steps in flight, do launches that take FEWER CU-microseconds (whole 256 x 256
# tiles on half the chip instead of the pair split; 256-row instead of 192-row tiles) beat the ones that are faster alone?
# Same box, alternating; headline step only.     tools/cutime_ab.sh [rounds]        (GPU box only)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
R=${1:-2}
for rnd in $(seq $R); do
  for arm in "0 2" "15=128 2" "15=256 2" "15=384 2" "0 3" "15=128 3" "15=384 3"; do
    set -- $arm
    t=""; [[ $1 != 0 ]] && t=$1
    KOSMOSX_TUNING="$t" python bench.py --steps 20 --warmup 5 --pipeline $2 --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tuning=${1} pipeline=${2}', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
