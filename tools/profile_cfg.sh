#!/bin/bash
# rocprofv3 --kernel-trace --stats of one other config's bench leg:  tools/profile_cfg.sh NAME <bench.py args>
#   -> gpurun_out/r05/prof_NAME/{kernel_stats.csv,line.json}
n=$1; shift
out=gpurun_out/r05/prof_$n
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_other_configs "$@" > $out/line.json 2> $out/err.log || exit 1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/stats
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/kernel_stats.csv")))
line = json.loads([l for l in open("$out/line.json") if l.startswith("{")][-1])
steps = None
tot = sum(float(r["TotalDurationNs"]) for r in rows if "_k<" in r["Name"] or "_k(" in r["Name"] or "gemm" in r["Name"])
print("$n", "%.2f ms/step" % line["ms_per_step"], "frac %.4f" % line["step_roofline"]["frac"])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%6d calls  avg %8.1f us  %5.1f %%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:100]))
PY
