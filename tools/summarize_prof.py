#!/usr/bin/env python
"""Condenses a gpurun_out/ directory written by scripts/gpu_full.sh into the small text/CSV files kept under
profiles/rNN/: rocprofv3 per-kernel stats, per-kernel averages of each PMC pass (with the gfx950 FETCH_SIZE x2
correction of MI355X_MICROARCH.md §HBM spelled out), the bench line, parity summaries."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
KEYS = ("gemm", "gemv", "quantize", "zero", "fillBuffer")

stats = os.path.join(src, "prof_stats", "r1_kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(dst, "rocprofv3_kernel_stats.csv"))

lines = []
for d in sorted(d_ for d_ in glob.glob(os.path.join(src, "pmc_*")) if os.path.isdir(d_)):
    f = os.path.join(d, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv"))):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines.append("## pass: rocprofv3 --kernel-trace --pmc %s -- python bench.py --steps 100 --warmup 10 --lean" % os.path.basename(d)[4:].replace("+", " "))
    for k, cs in agg.items():
        if not any(x in k for x in KEYS):
            continue
        n = len(next(iter(cs.values())))
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        extra = ""
        if "FETCH_SIZE" in avg:
            extra = "  -> HBM-side read bytes/launch = FETCH_SIZE[KB]*1024*2 (gfx950 x2 correction) = %.2f MB" % (avg["FETCH_SIZE"] * 1024 * 2 / 1e6)
        if "WRITE_SIZE" in avg:
            extra = "  -> write bytes/launch = WRITE_SIZE[KB]*1024 = %.2f MB (uncalibrated)" % (avg["WRITE_SIZE"] * 1024 / 1e6)
        lines.append("%-64s n=%-5d avg_dur_us(profiled)=%-8.2f %s%s" % (k[:64], n, sum(dur[k]) / max(len(dur[k]), 1), json.dumps({c: round(v, 1) for c, v in avg.items()}), extra))
    lines.append("")
open(os.path.join(dst, "pmc_summary.txt"), "w").write("\n".join(lines))

for name in ("bench.log", "gemm_bench.txt", "l2_stream.csv", "gpt2_parity.jsonl", "summary.txt", "nproc.txt", "rocminfo.txt", "t64_ablations.txt", "mfma_valu_probe.txt", "pmc_gemm_bench_sq.txt", "split_report.jsonl"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
p = os.path.join(src, "parity_report.jsonl")
if os.path.exists(p):
    worst = collections.defaultdict(float)
    for ln in open(p):
        j = json.loads(ln)
        key = (j.get("test"), j.get("type", "q4_K"))
        # (the FLASH_ATTN_EXT groups report under rel_l2_float64 / rel_l2_vs_f16_path: every rel_l2* key of a record counts — VERDICT r4 weak 3: the table printed 0.0 for them)
        vals = [v for kk, v in j.items() if kk.startswith("rel_l2") and isinstance(v, (int, float))]
        worst[key] = max([worst[key]] + vals)
    with open(os.path.join(dst, "parity_worst_rel_l2.txt"), "w") as f:
        f.write("worst rel-L2 vs the CPU oracle per (test group, weight type) over the -m gpu run\n")
        for k in sorted(worst, key=str):
            f.write("%-22s %-6s %.3e\n" % (k[0], k[1], worst[k]))
print("wrote", sorted(os.listdir(dst)))
