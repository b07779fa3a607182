#!/bin/bash
# fourth CU-time A/B (round 6): DYNAMIC tile walk of the persistent 256-column kernel (tuning key 15 & 512) under the throughput
# objective, alone and with the unsplit alternative to the pair split (15 & 128).   GPU box only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
for rnd in 1 2; do
  for arm in "|2" "15=512|2" "15=640|2" "15=512|3" "15=640|3" "15=512|1" "|1"; do
    IFS='|' read t P <<< "$arm"
    KOSMOSX_TUNING="$t" python bench.py --steps 20 --warmup 5 --pipeline $P --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tuning=[$t] pipeline=$P', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
