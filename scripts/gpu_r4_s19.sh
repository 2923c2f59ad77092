#!/bin/bash
# round 4, session 19: counters of the decode kernel at 4096 x 14336 (is it instructions or bytes?)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s19; mkdir -p $O
cd /tmp
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"; do
  name=$(echo "$pmc" | tr ' ' '+' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/$O/pmcd_$name" -o p -- python "$R/scripts/decode_loop.py" 4096 14336 200 > "$R/$O/pmcd_$name.log" 2>&1
done
cd "$R"
python3 - > $O/pmc_decode_counters.txt <<'PY'
import csv, glob, collections, os
print("# rocprofv3 --kernel-trace --pmc <pass> -- python scripts/decode_loop.py 4096 14336 200 ; per-launch averages of k_gemv_q_fused<Q4_K, 8, 2>")
for d in sorted(glob.glob('gpurun_out/s19/pmcd_*')):
    f=os.path.join(d,'p_counter_collection.csv')
    if not os.path.isdir(d) or not os.path.exists(f): continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for r in csv.DictReader(open(os.path.join(d,'p_kernel_trace.csv'))): dur[r['Kernel_Name']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,cs in agg.items():
        if 'gemv' not in k: continue
        print(k[:60], 'n=%d' % len(next(iter(cs.values()))), 'avg_dur_us(profiled)=%.2f' % (sum(dur[k])/max(1,len(dur[k]))), {c: round(sum(v)/len(v)) for c,v in cs.items()})
PY
rm -rf $O/pmcd_*/
cat $O/pmc_decode_counters.txt | cut -c1-500
