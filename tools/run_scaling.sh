#!/bin/bash
# Weak-scaling curve on one 8-GPU MI355X node: N = 1, 2, 4, 8 back to back (the contract's launch line); one JSON line each.
# usage: bash tools/run_scaling.sh [steps] [warmup]   -> gpurun_out/scale_N.json
STEPS=${1:-100}; WARM=${2:-5}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-pmc --no-cpu-baseline --no-extras | tail -1 > gpurun_out/scale_1.json
for N in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps $STEPS --warmup $WARM --no-extras | tail -1 > gpurun_out/scale_$N.json
done
python - <<'PY'
import json
base = None
for n in (1, 2, 4, 8):
    d = json.load(open(f"gpurun_out/scale_{n}.json"))
    base = base or d["value"]
    print(f"N={n}: {d['value']:.1f} event-frames/s  {d['ms_per_step']:.2f} ms/step  speed-up {d['value'] / base:.2f}x")
PY
