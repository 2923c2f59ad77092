#!/bin/bash
# round 3, session 3: tests touched since session 2, the determinism diagnosis, decode timings + a PMC pass of the decode kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; R=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi_ops.py tests/test_gpu_bench.py tests/test_gpu_widening.py -m gpu -q --tb=short -p no:cacheprovider -k "not flash_attn and not stock_harness" > gpurun_out/s3_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)" gpurun_out/s3_pytest.log | head; tail -2 gpurun_out/s3_pytest.log
timeout 600 python scripts/gpu_diag_determinism.py > gpurun_out/s3_determinism.log 2>&1; echo "determinism rc=$?"; grep "^{" gpurun_out/s3_determinism.log | cut -c1-420
for shape in "4096 4096" "4096 14336"; do timeout 200 python scripts/decode_loop.py $shape 400 2>&1 | tail -1; done
cd /tmp; export TMPDIR=/tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmcd_$name" -o p -- python "$R/scripts/decode_loop.py" 4096 14336 200 > "$R/gpurun_out/pmcd_$name.log" 2>&1
done
cd "$R"; python3 - <<'PY'
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
open('gpurun_out/pmc_decode_summary.txt','w').write("# rocprofv3 --kernel-trace --pmc <pass> -- python scripts/decode_loop.py 4096 14336 200  (FETCH_SIZE in KB: x1024 x2 on gfx950 = HBM-side read bytes)\n"+"\n".join(out)+"\n")
print("\n".join(out))
PY
rm -rf gpurun_out/pmcd_*/
