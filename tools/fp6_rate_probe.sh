#!/bin/bash
# What would fp6 (e2m3) correction products buy?  The side library built by `python kosmos-x_amd/build.py --fp6-rate-probe`
# issues the two correction MFMAs of every f16c kernel with the SAME registers declared e2m3 (the matrix pipe then runs them at
# the fp6 rate; the numbers are wrong, the loads / LDS traffic are today's 128-byte rows — an upper bound on the matrix-pipe
# side of the gain, without the 25 % smaller correction planes).  Same box, alternating, headline step + C3 f16c.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
P=$PWD/kosmos-x_amd/build/fp6probe/libkosmosx_hip_fp6probe.so
for rnd in 1 2; do
  for lib in default "$P"; do
    if [[ $lib == default ]]; then unset KOSMOSX_HIP_LIB; else export KOSMOSX_HIP_LIB=$lib; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g={(x['M'],x['N'],x['K']):x for x in d.get('gemm_shapes',[])}
print('$(basename $lib)', 'ms/step', d['ms_per_step'], 'samples/s', d['value'], ' '.join(f\"{k[1]}x{k[2]}:{v['ms_per_step']:.2f}ms\" for k,v in g.items() if k[0]==3648))"
  done
done
