#!/bin/bash
# t64 ablations + stage trace (timing only; ablated results are garbage)
mkdir -p gpurun_out; cd tools/microbench
B=40967; V="$B"; for a in 1 2 3 4 8 16 7 15 31 32 47; do V="$V,$((B + a*65536))"; done
GB_VARIANTS="$V" GB_SPLITKS=1 GB_TRACE_REPS=30 timeout 300 ./gemm_bench_abl 32768 8192 512 "$B,$((B + 15*65536))" > ../../gpurun_out/abl_t64_256_c5.txt 2>&1
B=24583; V="$B"; for a in 1 2 3 4 8 16 7 15 31 32; do V="$V,$((B + a*65536))"; done
GB_VARIANTS="$V" GB_SPLITKS=1 GB_TRACE_REPS=30 timeout 300 ./gemm_bench_abl 8192 8192 512 "$B,$((B + 15*65536))" > ../../gpurun_out/abl_t64_128_8k.txt 2>&1
timeout 120 ./mfma_valu 2000 > ../../gpurun_out/mfma_valu.txt 2>&1
cd ../..; cat gpurun_out/abl_t64_256_c5.txt; cat gpurun_out/abl_t64_128_8k.txt; head -8 gpurun_out/mfma_valu.txt
