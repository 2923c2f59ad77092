#!/bin/bash
mkdir -p gpurun_out; cd tools/microbench
B=40967
GB_VARIANTS="$B,$((B + 512*65536)),$((B + 3*65536)),$((B + 15*65536))" GB_SPLITKS=1 GB_TRACE_REPS=30 timeout 300 ./gemm_bench_abl 32768 8192 512 "$((B + 256*65536)),$((B + 768*65536))" > ../../gpurun_out/abl3_t64_256_c5.txt 2>&1
B=24583
GB_VARIANTS="4119,$B,$((B + 512*65536)),$((B + 3*65536))" GB_SPLITKS=2 GB_TRACE_SPLITK=2 timeout 300 ./gemm_bench_abl 4096 4096 512 "$B" > ../../gpurun_out/abl3_t64_128_head.txt 2>&1
GB_VARIANTS="4119,$B,$((B + 512*65536)),40967" GB_SPLITKS=1 timeout 300 ./gemm_bench_abl 8192 4096 512 "$((B + 256*65536)),$((B + 768*65536))" > ../../gpurun_out/abl3_t64_128_8k4k.txt 2>&1
GB_VARIANTS="4119,$B,$((B + 512*65536))" GB_SPLITKS=0 timeout 300 ./gemm_bench_abl 4096 11008 512 "$B" > ../../gpurun_out/abl3_c3.txt 2>&1
cd ../..; cat gpurun_out/abl3_t64_256_c5.txt gpurun_out/abl3_t64_128_head.txt gpurun_out/abl3_t64_128_8k4k.txt gpurun_out/abl3_c3.txt | grep -v "stage starts\|XCC_ID\|consumer\|producer\|^wave\|trace variant"
