#!/bin/bash
# round 3: rocprofv3 kernel stats + PMC passes of the FLASH_ATTN_EXT prefill loop (4096 x 4096, head size 128, 32 heads) -> gpurun_out/profile_fa/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/profile_fa; rm -rf gpurun_out/fa_stats gpurun_out/fa_pmc_*
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/fa_stats" -o f -- python "$R/scripts/fa_loop.py" 4096 4096 20 > "$R/gpurun_out/profile_fa/loop.log" 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/fa_pmc_$name" -o p -- python "$R/scripts/fa_loop.py" 4096 4096 10 > "$R/gpurun_out/fa_pmc_$name.log" 2>&1
done
cd "$R"
cp gpurun_out/fa_stats/*/f_kernel_stats.csv gpurun_out/profile_fa/rocprofv3_fattn_kernel_stats.csv 2>/dev/null || cp gpurun_out/fa_stats/f_kernel_stats.csv gpurun_out/profile_fa/rocprofv3_fattn_kernel_stats.csv
python3 - <<'PY'
import csv, glob, collections, os
out=[]
for d in sorted(glob.glob('gpurun_out/fa_pmc_*')):
    if not os.path.isdir(d): continue
    fs=glob.glob(os.path.join(d,'**','p_counter_collection.csv'), recursive=True)
    if not fs: continue
    f=fs[0]; t=os.path.join(os.path.dirname(f),'p_kernel_trace.csv')
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    if os.path.exists(t):
        for r in csv.DictReader(open(t)): dur[r['Kernel_Name']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,cs in agg.items():
        if 'flash_attn' not in k: continue
        out.append("%s | %s | n=%d avg_dur_us(profiled)=%.1f | %s" % (os.path.basename(d)[7:], k[:60], len(next(iter(cs.values()))), sum(dur[k])/max(1,len(dur[k])), {c: round(sum(v)/len(v),1) for c,v in cs.items()}))
open('gpurun_out/profile_fa/pmc_fattn_summary.txt','w').write("# rocprofv3 --kernel-trace --pmc <pass> -- python scripts/fa_loop.py 4096 4096 10  (FETCH_SIZE / WRITE_SIZE in KB: x1024 x2 on gfx950 = HBM-side bytes)\n"+"\n".join(out)+"\n")
print("\n".join(out))
PY
rm -rf gpurun_out/fa_stats gpurun_out/fa_pmc_*/
head -6 gpurun_out/profile_fa/rocprofv3_fattn_kernel_stats.csv; tail -1 gpurun_out/profile_fa/loop.log
