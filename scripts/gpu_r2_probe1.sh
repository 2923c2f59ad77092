#!/bin/bash
# round 2, GPU call 1: sanity of the cleaned default path, first run of k_gemm_kq_t64 (parity + timing beside the shipped kernels)
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "t64 or 64x128" 2>&1 | tail -15 ) > gpurun_out/r2_t64_tests.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "not 64x128" 2>&1 | tail -8 ) > gpurun_out/r2_parity_tests.txt
cd tools/microbench
for shape in "4096 4096 512" "4096 11008 512" "8192 4096 512" "4096 8192 512" "32768 8192 512" "8192 8192 512"; do
  GB_VARIANTS="0,4119,1031,24583,40967" GB_SPLITKS="0" GB_ROUNDS=4 timeout 240 ./gemm_bench $shape "" 2>&1 | grep -v "^  \|^trace\|^wave"
done > ../../gpurun_out/r2_gemm_bench1.txt 2>&1
cd ../..
tail -20 gpurun_out/r2_t64_tests.txt; tail -5 gpurun_out/r2_parity_tests.txt; cat gpurun_out/r2_gemm_bench1.txt
