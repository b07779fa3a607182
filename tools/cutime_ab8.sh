#!/bin/bash
# round 5's tower rules (256-column kernel for the CLIP tower's GEMMs) re-measured under the throughput objective: tuning key 15 bits
# 1 / 2 switch them off (the tower back on the 160 x 128 / 128 x 128 kernels, two workgroups per CU).   GPU box only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
for rnd in 1 2; do
  for t in "" "15=3" "15=1" "15=2"; do
    KOSMOSX_TUNING="$t" python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tuning=[$t] objective=throughput', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
