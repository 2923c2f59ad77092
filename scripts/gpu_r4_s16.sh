#!/bin/bash
# round 4, session 16: where the headline step's ~5 us beyond quantizer + GEMM go (kernel-trace timestamps of the step loop vs the single-kernel loops)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s16; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace" -o t -- python "$R/bench.py" --steps 300 --warmup 20 --lean > "$R/$O/trace.log" 2>&1
cd "$R"
python3 - > $O/step_gaps.txt <<'PY'
import csv, glob, statistics as st
f = glob.glob('gpurun_out/s16/trace/**/t_kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'gemm' if 'gemm' in r['Kernel_Name'] else ('quant' if 'quantize' in r['Kernel_Name'] else 'other')) for r in csv.DictReader(open(f))), key=lambda x: x[0])
seq = [r for r in rows if r[2] != 'other']
# classify each kernel by its predecessor: step loop = alternating quant / gemm; single-kernel loops = same kind back to back
out = {}
for a, b in zip(seq, seq[1:]):
    key = (a[2], b[2])
    out.setdefault(key, {'gap': [], 'dur_next': []})
    out[key]['gap'].append((b[0] - a[1]) / 1e3); out[key]['dur_next'].append((b[1] - b[0]) / 1e3)
print("# rocprofv3 --kernel-trace -- python bench.py --steps 300 --warmup 20 --lean ; pairs of consecutive launches (previous kernel, next kernel)")
for k, v in sorted(out.items()):
    g = [x for x in v['gap'] if x < 200]
    print("%-6s -> %-6s n=%-5d gap us: median %.2f mean %.2f   duration of the second us: median %.2f mean %.2f" % (k[0], k[1], len(g), st.median(g), st.mean(g), st.median(v['dur_next']), st.mean(v['dur_next'])))
PY
rm -rf $O/trace
cat $O/step_gaps.txt; tail -2 $O/trace.log | cut -c1-300
