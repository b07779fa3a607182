cd "${GRAFT_REPO_ROOT:-/root/repo}"
export KOSMOSX_NO_LOGGING_CONFIG=1
timeout 900 python -m pytest tests/test_f16c_gpu.py tests/test_pairk_gpu.py tests/test_lean_res_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2
for rnd in 1 2 3; do
  for lib in new old; do
    if [[ $lib == old ]]; then export KOSMOSX_HIP_LIB=$PWD/kosmos-x_amd/build/side/libkosmosx_hip_before_wscale.so; else unset KOSMOSX_HIP_LIB; fi
    python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'ms/step', d['ms_per_step'], 'samples/s', d['value'])"
  done
done
for lib in new old; do
  if [[ $lib == old ]]; then export KOSMOSX_HIP_LIB=$PWD/kosmos-x_amd/build/side/libkosmosx_hip_before_wscale.so; else unset KOSMOSX_HIP_LIB; fi
  python tools/bench_c3.py --precision f16c --steps 3 --warmup 1 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/$lib c3 /"
done
