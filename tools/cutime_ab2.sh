#!/bin/bash
# second CU-time A/B (round 6): one workgroup per tile (hardware dispatch = dynamic balance under co-scheduling, key 7 = -1)
# against the persistent walk, with and without the fewest-tiles rules.   GPU box only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
for rnd in 1 2; do
  for t in "" "7=-1" "15=256" "15=256,7=-1" "15=384" "15=384,7=-1"; do
    KOSMOSX_TUNING="$t" python bench.py --steps 20 --warmup 5 --pipeline 2 --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tuning=[$t] pipeline=2', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
