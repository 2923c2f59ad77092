#!/bin/bash
# round 3 profiles: rocprofv3 kernel stats + the PMC passes of bench.py --lean (headline), PMC + stats of the decode loop (4096 x 14336), gemm_bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -rf gpurun_out/prof_* gpurun_out/pmc_* gpurun_out/pmcd_*
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 200 --warmup 20 --lean > "$R/gpurun_out/rocprof_stats.log" 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmc_$name" -o p -- python "$R/bench.py" --steps 100 --warmup 10 --lean > "$R/gpurun_out/rocprof_pmc_$name.log" 2>&1
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_decode" -o d1 -- python "$R/scripts/decode_loop.py" 4096 14336 300 > "$R/gpurun_out/rocprof_decode_stats.log" 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmcd_$name" -o p -- python "$R/scripts/decode_loop.py" 4096 14336 200 > "$R/gpurun_out/pmcd_$name.log" 2>&1
done
cd "$R"; python tools/summarize_prof.py gpurun_out gpurun_out/profile_summary3 > gpurun_out/summarize.log 2>&1
cp gpurun_out/prof_decode/d1_kernel_stats.csv gpurun_out/profile_summary3/rocprofv3_decode_kernel_stats.csv 2>/dev/null
python3 - <<'PY'
import csv, glob, collections, os
out=[]
for d in sorted(glob.glob('gpurun_out/pmcd_*')):
    f=os.path.join(d,'p_counter_collection.csv')
    if not os.path.isdir(d) or not os.path.exists(f): continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for r in csv.DictReader(open(os.path.join(d,'p_kernel_trace.csv'))): dur[r['Kernel_Name']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,cs in agg.items():
        if 'gemv' not in k: continue
        out.append("%s | %s | n=%d avg_dur_us(profiled)=%.2f | %s" % (os.path.basename(d)[5:], k[:70], len(next(iter(cs.values()))), sum(dur[k])/max(1,len(dur[k])), {c: round(sum(v)/len(v),1) for c,v in cs.items()}))
open('gpurun_out/profile_summary3/pmc_decode_summary.txt','w').write("# rocprofv3 --kernel-trace --pmc <pass> -- python scripts/decode_loop.py 4096 14336 200  (FETCH_SIZE in KB: x1024 x2 on gfx950 = HBM-side read bytes)\n"+"\n".join(out)+"\n")
print("\n".join(out))
PY
rm -rf gpurun_out/prof_stats gpurun_out/prof_decode gpurun_out/pmc_*/ gpurun_out/pmcd_*/
head -8 gpurun_out/profile_summary3/rocprofv3_kernel_stats.csv; head -5 gpurun_out/profile_summary3/rocprofv3_decode_kernel_stats.csv; grep t64 gpurun_out/profile_summary3/pmc_summary.txt | cut -c1-300
for shape in "4096 4096 512" "4096 11008 512" "32768 8192 512" "8192 8192 512"; do GB_VARIANTS=0,4119 GB_SPLITKS=0 GB_ROUNDS=4 timeout 200 tools/microbench/gemm_bench $shape 2>&1 | grep -E "variant|M=" ; done | tee gpurun_out/profile_summary3/gemm_bench.txt
