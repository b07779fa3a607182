#!/bin/bash
# The 1 -> 8 GPU scaling curve of the headline metric in one command (run on an 8-GPU MI355X node; none has been
# available so far).  For every N in {1,2,4,8}: the bench with the direct (grouped send/recv) gather, with
# RCCL's all_gather, and without the gather — NCCL_DEBUG=INFO logs kept so the algorithm RCCL picked can be read off.
# Output: gpurun_out/scale/N<n>_<variant>.json (+ .log).  Usage: tools/scale_round.sh [steps] [warmup]
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 KOSMOSX_NO_LOGGING_CONFIG=1
STEPS="${1:-20}"; WARM="${2:-5}"
mkdir -p gpurun_out/scale
PORT=29600
for N in 1 2 4 8; do
  for V in direct all_gather nogather; do
    EXTRA="--gather-algo direct"; [[ $V == all_gather ]] && EXTRA="--gather-algo all_gather"; [[ $V == nogather ]] && EXTRA="--no-gather"
    [[ $N == 1 && $V != direct ]] && continue
    PORT=$((PORT + 1))
    if [[ $N == 1 ]]; then
      python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-extra > "gpurun_out/scale/N${N}_${V}.json" 2> "gpurun_out/scale/N${N}_${V}.log"
    else
      NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
        --master-port "$PORT" bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" --no-extra $EXTRA \
        > "gpurun_out/scale/N${N}_${V}.json" 2> "gpurun_out/scale/N${N}_${V}.log"
    fi
    tail -1 "gpurun_out/scale/N${N}_${V}.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=%d %-10s %10.1f samples/s  %.2f ms/step' % (d['n_gpus'], '$V', d['value'], d['ms_per_step']))" 2>/dev/null \
      || echo "N=$N $V failed: see gpurun_out/scale/N${N}_${V}.log"
  done
done
python - <<'PY'
import json, glob
base = None
for f in sorted(glob.glob("gpurun_out/scale/N*_direct.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    base = base or d["value"] / d["n_gpus"]
    print(f"N={d['n_gpus']}: {d['value']:.1f} samples/s, weak-scaling efficiency {d['value'] / d['n_gpus'] / base:.3f}")
PY
