#!/usr/bin/env python3
"""Where is the device idle? Reads a rocprofv3 --kernel-trace CSV of `bench.py` and prints, per optimizer step (delimited
by the adamw_k launches), the step's span, the sum of its kernel durations and the idle time between kernels, then the
largest gaps of one timed step.  usage: python tools/trace_gaps.py <..._kernel_trace.csv> [step index]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:56]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw_k" in r["Kernel_Name"]]
    ends = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] - i > 50]
    steps = list(zip(ends, ends[1:]))
    for si, (a, b) in enumerate(steps):
        seg = rows[a + 1:b + 1]
        t0, t1 = int(rows[a]["End_Timestamp"]), int(seg[-1]["End_Timestamp"])
        ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
        print(f"step {si:2d}: {len(seg)} kernels, span {(t1 - t0) / 1e6:7.3f} ms, kernels {ksum / 1e6:7.3f} ms, "
              f"idle {(t1 - t0 - ksum) / 1e3:7.0f} us")
    pick = int(sys.argv[2]) if len(sys.argv) > 2 else min(5, len(steps) - 1)
    a, b = steps[pick]
    seg = rows[a:b + 1]
    gaps = sorted((((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3, short(x["Kernel_Name"]),
                   short(y["Kernel_Name"])) for x, y in zip(seg, seg[1:])), reverse=True)
    print(f"largest gaps of step {pick} ({sum(1 for g in gaps if g[0] <= 0.01)} of {len(gaps)} are 0):")
    for g, x, y in gaps[:10]:
        print(f"  {g:8.1f} us  {x} -> {y}")


if __name__ == "__main__":
    main()
