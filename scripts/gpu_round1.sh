#!/bin/bash
# first GPU session: parity tests per group (separate processes so one fault does not hide the rest), smoke, bench, rocprof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/nproc.txt
for grp in "quantize" "golden" "gemv" "gemm_parity_auto" "gemm_variants or gemm_matches" "full_size" "mul_mat_id_parity"; do
  name=$(echo "$grp" | tr ' ' '_')
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "$grp" > gpurun_out/pytest_$name.log 2>&1
  echo "group [$grp] rc=$?" >> gpurun_out/summary.txt
  tail -3 gpurun_out/pytest_$name.log >> gpurun_out/summary.txt
done
timeout -k 10 900 python -m pytest tests/test_gpu_backend_plugin.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_plugin.log 2>&1
echo "plugin rc=$?" >> gpurun_out/summary.txt; tail -3 gpurun_out/pytest_plugin.log >> gpurun_out/summary.txt
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt
timeout -k 10 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/summary.txt
for v in 4 5 6 7; do for sk in 1 2 4; do
  timeout -k 10 120 python bench.py --steps 100 --warmup 10 --variant $v --splitk $sk --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('variant',$v,'splitk',$sk,'step_tflops',j['value'],'gemm_us',j['roofline']['us_per_launch'],'gemm_tflops',j['roofline']['achieved'])" >> gpurun_out/variants.txt 2>&1
done; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r1 -- python "$OLDPWD/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1
cd "$OLDPWD"; echo "rocprof rc=$?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt; cat gpurun_out/variants.txt; tail -2 gpurun_out/bench.log
