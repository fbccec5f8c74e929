#!/bin/bash
# training rate of the driver in its modes (synthetic Split-MNIST surrogate, first two tasks): bash tests/native/drv_modes.sh
for flags in "" "--graph" "--dataloader"; do
  rm -rf gpurun_out/drvm
  python experiments/vargp.py s-mnist --epochs 20 --M 100 --synthetic --n_synth 12000 --seed 1 $flags --log_dir gpurun_out/drvm > gpurun_out/drvm.log 2>&1 || tail -3 gpurun_out/drvm.log
  python - "$flags" <<PY
import json, sys
rows=[json.loads(l) for l in open("gpurun_out/drvm/scalars.jsonl")]
print("flags [%s]" % sys.argv[1], {r["key"]: round(r["value"],1) for r in rows if "steps_per_s" in r["key"]})
PY
done
