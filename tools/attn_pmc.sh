#!/bin/bash
# Instruction mix / busy counters of the attention kernels at one shape (default C4: 3 x 650) from separate --pmc passes of
# tools/attn_bench.py; prints per kernel: average per launch of each counter. usage: tools/attn_pmc.sh OUTDIR [SHAPE]
out=${1:-gpurun_out/attn_pmc}; shape=${2:-C4}
mkdir -p $out
export TMPDIR=/tmp AB_ONLY=$shape
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/attn_bench.py > $out/p$i.log 2>&1 || exit 1
done
python3 - $out <<'PY'
import csv, glob, os, sys, re
agg = {}
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if "attn" not in k:
            continue
        a = agg.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0, 0.0])
        a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for k, cs in sorted(agg.items()):
    print(k)
    for n, v in sorted(cs.items()):
        print(f"   {n:28s} {v[0] / v[1]:16.0f} per launch   (launch {v[2] / v[1]:7.1f} us, n={v[1]})")
PY
rm -rf $out/p*/*/*counter_collection.csv $out/p*/*/*kernel_trace.csv
