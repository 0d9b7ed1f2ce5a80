#!/bin/bash
# usage: tools/ab_bench.sh OUTDIR TAG_A LIB_A TAG_B LIB_B [rounds]   — interleaved same-box A/B of two library builds on
# the default bench (C2); LIB "-" = the in-tree libfvqa_hip.so. Prints one line per run.
out=$1; ta=$2; la=$3; tb=$4; lb=$5; n=${6:-3}
mkdir -p $out
for i in $(seq 1 $n); do
  for pair in "$ta:$la" "$tb:$lb"; do
    t=${pair%%:*}; l=${pair#*:}
    if [ "$l" = "-" ]; then unset FVQA_LIB; else export FVQA_LIB=$PWD/flipped-vqa_amd/fvqa/$l; fi
    python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | grep "^{" > $out/ab_${t}_$i.json
  done
done
unset FVQA_LIB
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/ab_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f, "%.3f ms  step %.4f  roof %.4f pass_ms %.3f nongemm %.3f %s"%(d["ms_per_step"], d["step_roofline"]["frac"], r["frac"], r["ms_per_step_this_pass"], r["non_gemm_ms_per_step"], d.get("invalid","")))
    except Exception as e: print(f, "ERR", e)
PY
