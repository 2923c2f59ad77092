#!/bin/bash
# quick GPU iteration: GEMM parity groups + variant sweep + gpt-2 node compare
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt gpurun_out/variants.txt
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm or full_size or golden" > gpurun_out/pytest_gemm.log 2>&1
echo "gemm parity rc=$?" >> gpurun_out/summary.txt; tail -5 gpurun_out/pytest_gemm.log >> gpurun_out/summary.txt
for cfg in "0 0" "5 1" "5 2" "7 1" "7 2" "7 4" "13 1"; do set -- $cfg
  timeout -k 10 120 python bench.py --steps 200 --warmup 20 --variant $1 --splitk $2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('variant',$1,'splitk',$2,'step_tflops',j['value'],'step_us',round(j['ms_per_step']*1e3,2),'gemm_us',j['roofline']['us_per_launch'],'gemm_tflops',j['roofline']['achieved'],'gemv_cold_us',j['decode']['us_per_gemv_cold_hbm'],'gemv_warm_us',j['decode']['us_per_gemv_cache_warm'])" >> gpurun_out/variants.txt 2>&1
done
CDNA4_GEMV_ROWS=2 timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('gemv_rows2 cold',j['decode']['us_per_gemv_cold_hbm'],'warm',j['decode']['us_per_gemv_cache_warm'])" >> gpurun_out/variants.txt 2>&1
# gpt-2 node-by-node compare
python tools/make_synth_gpt2.py /tmp/g_f32.bin > /dev/null && oracle/_ref/gpt-2-quantize /tmp/g_f32.bin /tmp/g_q4_0.bin q4_0 > /dev/null 2>&1
timeout -k 10 300 oracle/_ref/gpt2_harness /tmp/g_q4_0.bin CDNA40 ggml_amd/lib/libggml-cdna4.so COMPARE 8 1 16 > gpurun_out/gpt2_compare.log 2>&1
echo "gpt2 compare rc=$?" >> gpurun_out/summary.txt
grep -E "node|===" gpurun_out/gpt2_compare.log | awk '{ if ($0 ~ /rel_l2=/) { split($0,a,"rel_l2="); v=a[2]+0; if (v > 1e-5) print } else print }' | head -40 >> gpurun_out/summary.txt
cat gpurun_out/summary.txt; cat gpurun_out/variants.txt
