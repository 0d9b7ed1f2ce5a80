#!/bin/bash
# Interleaved A/B of two builds of the library on one box:  tools/ab_lib.sh VARIANT_LIB [PAIRS] [extra bench.py args]
# A = the product library, B = VARIANT_LIB (through FVQA_LIB). One C2 bench line per arm per pair -> gpurun_out/r05/ab_*.json
v=$1; n=${2:-3}; shift; shift
mkdir -p gpurun_out/r05
for i in $(seq 1 $n); do
  python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_other_configs "$@" > gpurun_out/r05/ab_A$i.json 2> gpurun_out/r05/ab_A$i.err || exit 1
  FVQA_LIB=$v python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_other_configs "$@" > gpurun_out/r05/ab_B$i.json 2> gpurun_out/r05/ab_B$i.err || exit 1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/ab_[AB]*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    sh = {k: round(v["avg_launch_us"], 1) for k, v in d["roofline"].get("per_instantiation", {}).items()}
    print(f[-8:-5], "%.3f ms  loss %s" % (d["ms_per_step"], d.get("loss", d.get("final_loss", ""))), sh)
PY
