#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt gpurun_out/variants.txt
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm_variants or full_size_prefill or gemm_parity_auto" > gpurun_out/pytest_variants.log 2>&1
echo "variants parity rc=$?" >> gpurun_out/summary.txt; tail -4 gpurun_out/pytest_variants.log >> gpurun_out/summary.txt
for cfg in "23 1" "23 2" "7 2" "7 1" "5 1" "5 2"; do set -- $cfg
  timeout -k 10 120 python bench.py --steps 200 --warmup 20 --variant $1 --splitk $2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('variant',$1,'splitk',$2,'step_tflops',j['value'],'step_us',round(j['ms_per_step']*1e3,2),'gemm_us',j['roofline']['us_per_launch'],'gemm_tflops',j['roofline']['achieved'])" >> gpurun_out/variants.txt 2>&1
done
R=$PWD; cd /tmp; export TMPDIR=/tmp
for cfg in "23 2" "7 2"; do set -- $cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_v$1_s$2" -o s -- python "$R/bench.py" --steps 50 --warmup 5 --variant $1 --splitk $2 --no-cpu-baseline > "$R/gpurun_out/rocprof_v$1_s$2.log" 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d "$R/gpurun_out/pmc_v$1_s$2" -o p -- python "$R/bench.py" --steps 20 --warmup 2 --variant $1 --splitk $2 --no-cpu-baseline > "$R/gpurun_out/rocprof_pmc_v$1_s$2.log" 2>&1
done
cd "$R"; cat gpurun_out/summary.txt; cat gpurun_out/variants.txt
for d in prof_v23_s2 prof_v7_s2; do python - "$d" <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/' + sys.argv[1] + '/s_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('gemm', 'zero')): print(sys.argv[1], r['Name'][:50], 'avg_us=%.2f' % (float(r['AverageNs'])/1e3))
PY
done
