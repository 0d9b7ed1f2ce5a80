out=gpurun_out/r3t/gemm_mix; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $out/p1 -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline > $out/p1.log 2>&1
python3 - $out <<'PY'
import csv, glob, os, sys, re
agg = {}
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if "gemm_sk" not in k and "attn" not in k and "rmsnorm" not in k:
            continue
        a = agg.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(agg.items()):
    v = {n: x[0] / x[1] for n, x in cs.items()}
    mf = max(v.get("SQ_INSTS_MFMA", 0), 1)
    print(f"{k[:70]:70s} n={list(cs.values())[0][1]:5d} VALU {v.get('SQ_INSTS_VALU',0)/1e6:7.2f}M MFMA {v.get('SQ_INSTS_MFMA',0)/1e6:6.2f}M  (VALU-MFMA)/MFMA {(v.get('SQ_INSTS_VALU',0)-mf)/mf:5.2f}  LDS {v.get('SQ_INSTS_LDS',0)/1e6:6.2f}M SALU {v.get('SQ_INSTS_SALU',0)/1e6:6.2f}M VMEM {v.get('SQ_INSTS_VMEM',0)/1e6:6.2f}M")
PY
rm -rf $out/p1/*/*counter_collection.csv $out/p1/*/*kernel_trace.csv
