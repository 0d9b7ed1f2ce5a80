#!/bin/bash
# Interleaved A/B of one environment switch on one box:  tools/ab_env.sh "VAR=value" [PAIRS] [extra bench.py args]
# A = as shipped, B = with VAR=value exported. One C2 bench line per arm per pair -> gpurun_out/r05/abe_*.json
v=$1; n=${2:-3}; shift; shift
mkdir -p gpurun_out/r05
for i in $(seq 1 $n); do
  python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_other_configs "$@" > gpurun_out/r05/abe_A$i.json 2> gpurun_out/r05/abe_A$i.err || exit 1
  env "$v" python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_other_configs "$@" > gpurun_out/r05/abe_B$i.json 2> gpurun_out/r05/abe_B$i.err || exit 1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/abe_[AB]*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    sr = d["step_roofline"]
    print(f[-8:-5], "%.3f ms %.1f samples/s loss %.6f  frac %.4f executed %s  gemm frac %.4f non-gemm %.3f ms" % (
        d["ms_per_step"], d["value"], d["loss"], sr["frac"], ("%.4f" % sr["frac_executed"]) if "frac_executed" in sr else "-",
        d["roofline"]["frac"], d["roofline"]["non_gemm_ms_per_step"]),
        {k: round(v["avg_launch_us"], 1) for k, v in d["roofline"]["per_instantiation"].items()})
PY
