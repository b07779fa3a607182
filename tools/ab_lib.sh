#!/bin/bash
# Same-box alternating A/B of the headline step between the shipped library and a side library (KOSMOSX_HIP_LIB):
#   tools/ab_lib.sh <side .so> [rounds]      (GPU box only)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
SIDE=$1; R=${2:-3}
for rnd in $(seq $R); do
  for lib in shipped "$SIDE"; do
    if [[ $lib == shipped ]]; then unset KOSMOSX_HIP_LIB; else export KOSMOSX_HIP_LIB=$PWD/$lib; fi
    python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$(basename $lib)', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
