#!/bin/bash
# third CU-time A/B (round 6): the throughput objective with smaller persistent grids (a launch leaves CUs to the other stream
# by construction) and more steps in flight.   GPU box only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
for rnd in 1 2; do
  for arm in "|2|latency" "|2|auto" "7=128|2|auto" "7=192|2|auto" "7=128|3|auto" "7=128|4|auto" "7=224|2|auto"; do
    IFS='|' read t P obj <<< "$arm"
    KOSMOSX_TUNING="$t" python bench.py --steps 20 --warmup 5 --pipeline $P --objective $obj --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tuning=[$t] pipeline=$P objective=$obj', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
