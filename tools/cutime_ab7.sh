#!/bin/bash
# (A/B bits: rule excluded for the XPos launches = 15 & 1024, for the others = 15 & 2048)
# which launches carry the throughput objective's gain (round 6): A/B bits 15 & 1024 (the decoder's XPos qkv launches keep 192-row
# tiles and their lean epilogue) and 15 & 2048 (only they take 256-row tiles).   GPU box only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
for rnd in 1 2 3; do
  for t in "" "15=1024" "15=2048"; do
    KOSMOSX_TUNING="$t" python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tuning=[$t] objective=throughput', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
