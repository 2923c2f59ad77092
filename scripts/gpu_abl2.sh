#!/bin/bash
mkdir -p gpurun_out; cd tools/microbench
L="1 2 3 4 8 15 16 32 64 128 192"
B=40967; V="$B"; for a in $L; do V="$V,$((B + a*65536))"; done
GB_VARIANTS="$V" GB_SPLITKS=1 GB_TRACE_REPS=30 timeout 300 ./gemm_bench_abl 32768 8192 512 "$((B + 256*65536)),$((B + 271*65536))" > ../../gpurun_out/abl2_t64_256_c5.txt 2>&1
B=24583; V="$B"; for a in $L; do V="$V,$((B + a*65536))"; done
GB_VARIANTS="4119,$V" GB_SPLITKS=2 GB_TRACE_SPLITK=2 GB_TRACE_REPS=30 timeout 300 ./gemm_bench_abl 4096 4096 512 "$((B + 256*65536))" > ../../gpurun_out/abl2_t64_128_head.txt 2>&1
GB_VARIANTS="4119,$V" GB_SPLITKS=1 timeout 300 ./gemm_bench_abl 8192 4096 512 "$((B + 256*65536))" > ../../gpurun_out/abl2_t64_128_8k4k.txt 2>&1
cd ../..; cat gpurun_out/abl2_t64_256_c5.txt gpurun_out/abl2_t64_128_head.txt gpurun_out/abl2_t64_128_8k4k.txt | grep -v "stage starts"
