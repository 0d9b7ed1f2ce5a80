#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of `bench.py` per kernel (profiles/r02_pmc_*.json).

usage: python tools/pmc_summary.py OUT.json DIR [DIR ...]
Each DIR holds one pass (its *_counter_collection.csv; rocprofv3 writes one row per dispatch and counter). Counters
are averaged per launch over every dispatch of a kernel; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests of
wide coalesced reads at 64 bytes: /opt/skills/guides/MI355X_MICROARCH.md, HBM) and both TCC sizes are KiB. Derived,
where a pass holds the inputs:
  mfma_busy_frac   SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 4 SIMDs * CUs) — matrix-pipe busy share of the
                   launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs; a busy cycle is counted per SIMD)
  lds_conflict_frac SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_frac        SQ_WAIT_ANY / SQ_WAVE_CYCLES; issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
The shader clock is NOT derived here: GRBM_GUI_ACTIVE / 8 / duration reads high on launches shorter than ~0.3 ms (it gave
4.8 GHz for the RMSNorm kernels of a 2.4 GHz part); the in-loop clock comes from s_memtime / s_memrealtime stamps of the
diagnostic build, tools/sk_clock.py -> profiles/r04_clock.log.
"""
import csv
import glob
import json
import os
import re
import sys

KEEP = ("gemm_sk_256", "gemm4w_sk_k", "fvqa_g4::gemm4w_k", "fewrows_partial_k", "fewrows_finish_k", "attn_fwd_mfma_k", "attn_bwd_fused_k",
        "rmsnorm_fwd_k", "rmsnorm_bwd_k", "gemm_nt_skinny")
N_CU = 256


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name).replace("__hip_bfloat16", "bf16")


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k.startswith(KEEP):
                    continue
                a = agg.setdefault(k, {})
                c = a.setdefault(r["Counter_Name"], [0.0, 0, 0.0])
                c[0] += float(r["Counter_Value"])
                c[1] += 1
                c[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    res = {}
    for k, cs in sorted(agg.items()):
        e = {"launches_sampled": max(v[1] for v in cs.values())}
        avg = {n: v[0] / v[1] for n, v in cs.items()}
        dur = {n: v[2] / v[1] for n, v in cs.items()}
        for n, v in avg.items():
            if n == "FETCH_SIZE":
                e["fetch_bytes_per_launch_corrected"] = v * 1024 * 2
            elif n == "WRITE_SIZE":
                e["write_bytes_per_launch"] = v * 1024
            else:
                e[n + "_per_launch"] = v
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch_corrected"] + e["write_bytes_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg:
            e["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8 * 4 * N_CU)
            e["avg_duration_us_under_pmc"] = dur["GRBM_GUI_ACTIVE"]
        if "SQ_LDS_BANK_CONFLICT" in avg and "SQ_LDS_IDX_ACTIVE" in avg and avg["SQ_LDS_IDX_ACTIVE"] > 0:
            e["lds_conflict_frac"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
        if "SQ_WAVE_CYCLES" in avg and avg["SQ_WAVE_CYCLES"] > 0:
            for src, dst in (("SQ_WAIT_ANY", "wait_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"),
                             ("SQ_ACTIVE_INST_ANY", "active_frac")):
                if src in avg:
                    e[dst] = avg[src] / avg["SQ_WAVE_CYCLES"]
        res[k] = e
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "flipped-vqa_amd"))
    from fvqa import _lib, build                            # host-only calls: ABI version + hash of the kernel sources
    # run this in the SAME gpurun call as the counter passes: the tag is the code the counters were collected on
    doc = {"fvqa_version": int(_lib.load().fvqa_version()), "source_hash": build.source_hash(),
           "command": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python3 bench.py --steps 2 --warmup 1 "
                      "--no_cpu_baseline (C2: LLaMA-7B bf16 B=8 S=128 VQA)",
           "kernels": res}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
