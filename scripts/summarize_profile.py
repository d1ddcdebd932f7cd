#!/usr/bin/env python3
"""Condense rocprofv3 output directories (kernel-trace --stats, and separate --pmc passes) into one text summary.
usage: summarize_profile.py OUT.txt STATS_DIR [PMC_DIR ...]"""
import collections
import csv
import glob
import os
import sys

out, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
lines = []

# kernel -> the category bench.py times it under (ILQR_PROF_*); bench.pmc_traffic reads the pmc[category] lines
CATEGORY = (("k_backward", "backward"), ("k_forward_wg", "forward"), ("k_forward_mfma", "forward"), ("k_forward_lin", "forward"),
            ("k_select", "forward"), ("k_cpl_linearize", "backward"), ("k_cp_solve", "backward"), ("k_cp_linearize", "backward"),
            ("k_cpl_linesearch", "forward"), ("k_bt_linesearch", "forward"), ("k_cpl_states", "rollout"), ("k_cpl_quad", "rollout"), ("k_cp_init", "rollout"),
            ("k_init_", "rollout"), ("k_cpl_final", "apply"), ("k_cp_final", "apply"), ("k_apply", "apply"), ("k_blend", "apply"))


def category(name):
    for key, cat in CATEGORY:
        if key in name:
            return cat
    return None

for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
    lines.append(f"# rocprofv3 --kernel-trace --stats : {os.path.relpath(f)}")
    lines.append(f"{'kernel':<70} {'calls':>6} {'avg_us':>12} {'total_ms':>10} {'pct':>6}")
    for r in csv.DictReader(open(f)):
        lines.append(f"{r['Name'][:70]:<70} {r['Calls']:>6} {float(r['AverageNs'])/1e3:>12.1f} {float(r['TotalDurationNs'])/1e6:>10.2f} {float(r['Percentage']):>6.2f}")
for d in pmc_dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        meta = {}
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:70], r["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
            meta[r["Kernel_Name"][:70]] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
        lines.append(f"# rocprofv3 --pmc : {os.path.relpath(f)}  (per-dispatch average; FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts half of a coalesced stream)")
        cat = collections.defaultdict(float)
        for (k, c), (n, s) in sorted(agg.items()):
            v, a, sg, sc, lds = meta[k]
            lines.append(f"{k:<70} {c:<12} n={n:<4} avg={s/n:>14.1f}   vgpr={v} agpr={a} sgpr={sg} scratch={sc} lds={lds}")
            if category(k) and "k_select" not in k:  # per launch of the category's main kernel(s); the decision kernel moves kilobytes
                cat[(category(k), c)] += s / n
        for (ct, c), v in sorted(cat.items()):
            lines.append(f"pmc[{ct}] {c} avg={v:.1f}   (KiB per launch of the category, summed over its kernels)")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
