#!/bin/bash
# One default bench line (C2 + the other-config legs, no CPU baseline) tagged with the box it ran on:  tools/bench_box.sh TAG
# -> gpurun_out/r05/box_TAG.json (the line) + box_TAG.txt (GPU unique id). Run once per gpurun call: every call gets a fresh box.
t=${1:-x}; mkdir -p gpurun_out/r05
(hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id") > gpurun_out/r05/box_$t.txt
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r05/box_$t.json 2> gpurun_out/r05/box_$t.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r05/box_$t.json") if l.startswith("{")][-1])
print("$t", open("gpurun_out/r05/box_$t.txt").read().split()[-1], "%.1f samples/s %.3f ms step %.4f gemm %.4f" % (d["value"], d["ms_per_step"], d["step_roofline"]["frac"], d["roofline"]["frac"]),
      {k: round(v["samples_per_s"], 1) for k, v in d.get("other_configs", {}).items()}, d.get("invalid", ""))
PY
