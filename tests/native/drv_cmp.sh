for mode in 1 0; do
rm -rf gpurun_out/drv$mode
VARGP_EPOCH_GRAPHS=$mode python experiments/vargp.py s-mnist --epochs 150 --M 60 --graph --synthetic --n_synth 12000 --seed 1 --log_dir gpurun_out/drv$mode > gpurun_out/drv$mode.log 2>&1; echo mode=$mode rc=$?
python - <<PY
import json
rows=[json.loads(l) for l in open("gpurun_out/drv$mode/scalars.jsonl")]
print({r["key"]: round(r["value"],3) for r in rows if "steps_per_s" in r["key"] or ("acc_best" in r["key"] and "train" in r["key"])})
PY
done
