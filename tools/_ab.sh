set -e
mkdir -p gpurun_out/r3
for i in 1 2 3; do
FVQA_OLD_ZERO=1 timeout -k 10 200 python bench.py --no_cpu_baseline --steps 20 2>/dev/null | grep "^{" > gpurun_out/r3/z_old_$i.json
timeout -k 10 200 python bench.py --no_cpu_baseline --steps 20 2>/dev/null | grep "^{" > gpurun_out/r3/z_new_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/z_*_?.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), round(d['value'],1))
PY
