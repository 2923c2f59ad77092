#!/bin/bash
# kernel-level profile of gpt-2 117M (synthetic weights, q4_0) decode on the plug-in: which kernels, how many per token, how long
R="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$R/gpurun_out"; cd /tmp; export TMPDIR=/tmp
python "$R/tools/make_synth_gpt2.py" /tmp/f32.bin > /dev/null && "$R/oracle/_ref/gpt-2-quantize" /tmp/f32.bin /tmp/q4_0.bin q4_0 > /dev/null 2>&1
for np in 8 200; do
  "$R/oracle/_ref/gpt2_harness" /tmp/q4_0.bin CDNA40 "$R/ggml_amd/lib/libggml-cdna4.so" /tmp/o.bin $np 32 16 | tail -1
done
"$R/oracle/_ref/gpt2_harness" /tmp/q4_0.bin CPU - /tmp/o.bin 200 16 16 | tail -1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/gpt2_prof" -o g -- "$R/oracle/_ref/gpt2_harness" /tmp/q4_0.bin CDNA40 "$R/ggml_amd/lib/libggml-cdna4.so" /tmp/o.bin 200 32 16 > "$R/gpurun_out/gpt2_prof.log" 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/gpt2_prof/g_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms", tot/1e6, "calls", sum(int(r["Calls"]) for r in rows))
for r in rows[:25]: print("%-90s calls %6s avg %8.1f ns  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
