#!/bin/bash
# Profiles of the default bench (C2) on the MI355X box; everything lands under $1 (default gpurun_out/r5prof).
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 5` (per-kernel durations + the bench's own line)
#   2. four SEPARATE --pmc passes of `bench.py --steps 2 --warmup 1` (counters never share a run with the stats trace)
#   3. tools/pmc_summary.py over the passes, tagged with the hash of the kernel sources they ran on
out=${1:-gpurun_out/r5prof}
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_other_configs > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err || exit 1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_other_configs > $out/pmc$i.json 2> $out/pmc$i.err || exit 1
done
python3 tools/pmc_summary.py $out/pmc_summary.json $out/pmc1 $out/pmc2 $out/pmc3 $out/pmc4 > /dev/null
f=$(ls $out/stats/*/*kernel_stats.csv | head -1)
cp $f $out/kernel_stats.csv
#   4. the GEMM roofline fraction on rocprof's clock beside the line's own (same run): frac_rocprof, sum of kernel time / step
python3 tools/rocprof_frac.py $out/kernel_stats.csv $out/bench_under_rocprof.json $out/rocprof_frac.json > /dev/null
python3 tools/trace_gaps.py $(ls $out/stats/*/*kernel_trace.csv | head -1) 10 > $out/trace_gaps.log 2>&1
rm -rf $out/stats/*/*kernel_trace.csv $out/pmc*/*/*counter_collection.csv $out/pmc*/*/*kernel_trace.csv   # (large; summaries kept)
ls $out
