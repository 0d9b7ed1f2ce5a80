#!/usr/bin/env python3
"""avg duration (us) of the kernels whose name contains one of the given substrings, from a rocprofv3 kernel_stats.csv:
   python tools/kernel_times.py <kernel_stats.csv> attn_bwd_fused attn_fwd_mfma rmsnorm"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
for key in sys.argv[2:]:
    for r in rows:
        if key in r["Name"]:
            print(f"{key:18s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:8.2f} us   {r['Name'][:70]}")
